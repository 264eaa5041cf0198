for r in 1 2; do for m in 0 1 2 3 4 8; do CURV_ALT_LIB=tools/micro/libcurv_flat_ab$m.so python tools/ab_update.py 2>&1 | grep update; done; done
