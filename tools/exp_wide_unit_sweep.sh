for F in 16 64 128 256; do for u in 1 2 4; do echo "F=$F unit=$u: $(MI355_WIDE_UNIT=$u timeout 300 python tools/wide_times.py $F 10 2>&1 | grep noise)"; done; done
