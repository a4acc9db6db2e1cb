for g in 0 16 32; do echo "GRID $g"; BWAMS_ERT_GRID=$g timeout -k 10 300 python tools/ert_scale.py 2>&1 | grep "^ERT  ms" | tail -1; done
