for m in ${MODES:-0 1 2}; do echo "== BTS_IGEMM_DBG=$m"; BTS_IGEMM_DBG=$m python scripts/conv_microbench.py 5 2>&1 | grep -E "128\^3 32->32|dec.L0 conv1|64\^3 64->64" | cut -c1-84; done
