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
# interference check (wgrad_stream.h): every variant must stay bit-identical to its solo run while a
# register-only MFMA loop of another stream shares its SIMDs.  wgrad_test = the library's flags (no packed-fp32 VALU ops);
# wgrad_test_slp = the same source with the SLP vectoriser on (v_pk_mul_f32 in the mask multiplications): differs every run.
for shape in "1 80000 5 28 24 256" "1 80000 5 28 48 256" "1 80000 5 68 64 256" "1 80000 5 28 128 256" "1 80000 5 100 32 256"; do
  echo -n "no packed fp32, $shape: "; MFMAN=1024 CONC=12 NROT=1 ./wgrad_test $shape 2 | grep "CONC:"
done
if [ -x ./wgrad_test_slp ]; then echo -n "packed fp32 (SLP on), 1 80000 5 28 48 256: "; MFMAN=1024 CONC=12 NROT=1 ./wgrad_test_slp 1 80000 5 28 48 256 2 | grep "CONC:"; fi
