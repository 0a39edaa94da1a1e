#!/bin/bash
# usage (GPU box, repo root): bash tools/kbench/ab_bf16.sh OLD_BINARY NEW_BINARY > gpurun_out/ab.txt
# correctness of NEW on odd shapes for every tile configuration, then OLD vs NEW timing on the model's shapes
cd tools/kbench
OLD=$1; NEW=$2
echo "== correctness ($NEW)"
for cfg in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  ./$NEW 300 200 88 3 nt $cfg 2 3 | tr '\n' ' '; echo
  ./$NEW 333 136 77 2 nn $cfg 1 3 1 | tr '\n' ' '; echo
  ./$NEW 700 328 200 2 nn $cfg 3 3 | tr '\n' ' '; echo
done
echo "== timing"
while read -r sh; do
  [ -z "$sh" ] && continue
  for b in $OLD $NEW $OLD $NEW; do printf "%-10s %-34s" $b "$sh"; CB_ONLY=${CBO:-} ./$b $sh | tail -1; done
done <<'LIST'
7372 2048 1843 1 nn 4 1 30
7372 2048 1843 1 nn 3 1 30
7372 2048 1843 1 nn 2 1 30
7372 1024 1843 1 nn 1 1 30
7372 1024 1843 1 nn 9 1 30
7372 1024 1843 1 nn 13 1 30
1843 2048 1843 4 nn 1 4 30
1843 2048 1843 4 nn 1 2 30
1843 1024 1843 4 nn 1 4 30
1843 1843 1843 1 nn 1 2 30
7372 1843 2176 12 nt 4 1 5
7372 1843 1024 12 nt 3 1 5
32768 4224 8192 1 nn 4 1 3
LIST
