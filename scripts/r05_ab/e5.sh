#!/bin/bash
cd "$GRAFT_REPO_ROOT"
L=$GRAFT_REPO_ROOT/3d-brain-tumor-segmentation_amd/csrc/build/libbts_hip_dbg.so
for dbg in 0 1 2 4 8 16 3 5 7 31; do BTS_HIP_LIB=$L BTS_S1D_DBG=$dbg python scripts/lp_s1d_inf_exp.py 2>/dev/null; done
