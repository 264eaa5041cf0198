for r in 1 2 3 4 5 6; do for v in 0 1; do echo -n "r$r fork_after_near=$v: "; CURV_FORK_AFTER_NEAR=$v python bench.py --stream-probe none --batch 32 2>/dev/null | tail -1; done; done
