for i in 1 2; do for v in "$@"; do MCR_DEV_LIB=$v VARIANTS=6 python tools/time_local_pct_ab.py 2>&1 | tail -1; done; done
