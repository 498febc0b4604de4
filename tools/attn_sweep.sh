for f in 1 0; do for w in 5 4 3; do echo "fused=$f waves=$w"; TIMHIP_ATTN_FUSED=$f TIMHIP_ATTN_WAVES=$w python tools/attn_one.py 0.1 2>&1 | grep -E "fwd|bwd"; done; done
