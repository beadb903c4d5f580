#!/bin/bash
# last measurement session of round 4 (library with the per-unit back-end flags, EXPERIMENTS R4-10): GPU suite, the driver's
# default line + kernel stats + smoke, the five workload lines, full-film parity of all five configurations
cd /root/repo; export TMPDIR=/tmp; OUT=gpurun_out/r04c; mkdir -p $OUT
bash tools/r04_verify_session.sh; cp -r gpurun_out/r04b/* $OUT/
for w in cornell dragon matpreview-rc matpreview-rd volumetric; do
  timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-also > $OUT/bench_$w.line 2> $OUT/bench_$w.err
  echo "bench $w rc=$?"; cp gpurun_out/bench_detail_${w}_n1.json $OUT/bench_$w.json; head -c 260 $OUT/bench_$w.line; echo
done
timeout 1200 python tests/full_size_parity.py > $OUT/full_size_parity.log 2>&1; echo "full parity rc=$?"; cp gpurun_out/full_size_parity.json $OUT/full_size_parity.json
grep -o '"config": "[^"]*"\|"frac_bit_exact": [0-9.]*\|"hip_msamples_per_s": [0-9.]*' $OUT/full_size_parity.log | paste - - - | cut -c1-200
