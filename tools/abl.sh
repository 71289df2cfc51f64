for d in 0 1 2 14 15; do echo "dbg=$d"; N3D_CONV_DBG=$d python tools/dbg_race.py mode0_big,flat4 2>&1 | tail -4; done
