#!/bin/bash
# streaming weight gradient: METR-LA / EXPY-TKY / PEMS-BAY shapes and ragged ones
cd tools/kbench
./wgrad_test 12 13248 5 68 128 21 10     # METR-LA encoder gate
./wgrad_test 12 13248 5 68 64 21 10      # METR-LA encoder update
./wgrad_test 12 13248 5 132 128 21 10    # METR-LA decoder gate
./wgrad_test 12 13248 5 132 256 21 10    # METR-LA decoder gate (O = 2 (H + D))
./wgrad_test 12 13248 5 132 128 21 10
./wgrad_test 6 58976 5 36 64 42 10       # EXPY-TKY encoder gate
./wgrad_test 6 58976 5 36 32 42 10
./wgrad_test 6 58976 5 68 64 42 10
./wgrad_test 12 20800 5 68 128 21 10     # PEMS-BAY
./wgrad_test 3 621 5 16 24 4 3
./wgrad_test 3 621 5 84 136 4 3          # two row blocks x two column blocks, ragged
./wgrad_test 2 250 3 20 8 3 3
./wgrad_test 3 900 5 24 36 85 3          # more chunks than 32-row blocks
./wgrad_test 12 13248 5 68 128 42 10
./wgrad_test 12 13248 5 68 128 10 10
# co-residency check (see wgrad_stream.h): the launch must stay bit-identical while a 64 KB-LDS GEMM runs on another stream
DCFG=0 CONC=30 NROT=1 ./wgrad_test 1 80000 5 28 48 256 2 | grep "CONC:"
