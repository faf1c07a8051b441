#include "common.h"
#include "bts_internal.h"
extern "C" const char* bts_version(void) { return "bts_hip 0.1 gfx950"; }
