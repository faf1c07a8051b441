// Internal constants shared by the .hip translation units; the public C ABI is include/bts_hip.h.
#pragma once
#include "../../include/bts_hip.h"
