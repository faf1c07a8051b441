for m in 0 1 2; do echo "== BTS_WGRAD_DBG=$m"; BTS_WGRAD_DBG=$m python scripts/conv_microbench.py 5 2>&1 | grep -E "128\^3 32->32|dec.L0 conv1|64\^3 64->64|128\^3 ptwise" | cut -c1-46,84-; done
