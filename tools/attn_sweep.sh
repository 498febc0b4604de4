for a in 0 1 2 4 6 7; do echo "ATT_ABL=$a"; ATT_ABL=$a python tools/attn_one.py 0.1 2>&1 | grep -E "bwd"; ATT_ABL=$a python tools/attn_one.py 0.0 2>&1 | grep -E "bwd"; done
