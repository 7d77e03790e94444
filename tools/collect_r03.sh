#!/bin/bash
# Round-3 measurement batch on the GPU box: everything DESIGN.md section 7 (round 3) quotes.  Writes gpurun_out/r03/.
# Usage: tools/collect_r03.sh [part ...]   parts: bench trace sq soak hmc map config5 extra fuzz   (default: all but extra and fuzz)
# (extra: the sweeps / probes of the second half of the round; the tools/ubench binaries are built with the hipcc lines in their sources)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/r03
mkdir -p $OUT
PARTS="${*:-bench trace sq soak hmc map config5}"
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has bench; then
  python bench.py > $OUT/bench_default_line.json 2> $OUT/bench_default.err
  python bench.py --no-cpu-baseline --phase-profile > /dev/null 2> $OUT/phase_profile.txt
  for cfg in "1 4" "32 8" "128 8" "256 8"; do set -- $cfg
    python bench.py --no-cpu-baseline --spectra $1 --chains $2 --steps 5 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('units %d: %.4g evals/s, %.2f us per round' % (d['config']['units_per_gpu'], d['value'], d['ms_per_step']))"
  done > $OUT/few_chain_points.txt 2>&1
  python tools/bench_logp.py > $OUT/logp_micro.txt 2>&1
fi
if has trace; then
  bash tools/profile_bench.sh r03 > $OUT/profile_bench.log 2>&1
  for f in bench_line.json kernel_stats.txt pmc_FETCH_SIZE.txt pmc_WRITE_SIZE.txt pmc_traffic.json; do cp gpurun_out/prof_r03/$f $OUT/ 2>/dev/null; done
fi
if has sq; then
  bash tools/profile_sq.sh r03 > $OUT/sq.log 2>&1
  cp gpurun_out/sq_r03/summary.txt $OUT/sq_counters_nuts_kernel.txt 2>/dev/null
fi
if has soak; then
  python tools/soak_sampling.py 512 8 500 500 > $OUT/config4_end_to_end.txt 2>&1
  python tools/soak_sampling.py 1536 8 200 200 > $OUT/oversubscribed.txt 2>&1
  echo "---- the same run with the layout frozen (BDRT_COMPACTION=0) ----" >> $OUT/oversubscribed.txt
  BDRT_COMPACTION=0 python tools/soak_sampling.py 1536 8 200 200 >> $OUT/oversubscribed.txt 2>&1
  python tools/config3_run.py > $OUT/config3.txt 2>&1
fi
if has hmc; then
  python tools/hmc_suite_run.py > $OUT/hmc_suite.txt 2>&1
fi
if has map; then
  python tools/map_timing.py > $OUT/map_timing.txt 2>&1
  python tools/map_kat_table.py > $OUT/map_kats.txt 2>&1
  python tools/map_suite_run.py > $OUT/map_suite.txt 2>&1
  python tools/map_batch_timing.py > $OUT/map_batch_timing.txt 2>&1
  python tools/ridge_timing.py > $OUT/ridge_timing.txt 2>&1
fi
if has config5; then
  python tools/bench_config5.py 4096 --phase-profile > $OUT/config5.txt 2>&1
  python tools/bench_config5.py 4096 --series-outliers >> $OUT/config5.txt 2>&1
  python tools/config5_run.py > $OUT/config5_run.txt 2>&1
fi
if has extra; then
  python tools/few_points_sweep.py > $OUT/few_points_sweep.txt 2>&1
  python tools/hmc_suite2_run.py > $OUT/hmc_suite2.txt 2>&1
  python tools/solo_duo_probe.py > $OUT/solo_duo.txt 2>&1
  python tools/soak_sampling.py 128 8 500 500 > $OUT/soak_128x8.txt 2>&1
  python tools/scatter_2rc.py > $OUT/scatter_2rc.txt 2>&1
  (for b in f64_overlap toep_loop toep_gemm_probe; do echo "== tools/ubench/$b"; ./tools/ubench/$b; echo; done) > $OUT/ubench.txt 2>&1
fi
if has fuzz; then
  python -m tests.fuzz_parity --first 3000 --count 400 > $OUT/fuzz_parity.txt 2>&1
  python -m tests.fuzz_ridge --first 3000 --count 400 > $OUT/fuzz_ridge.txt 2>&1
  python -m tests.fuzz_inverter --first 3000 --count 200 > $OUT/fuzz_inverter.txt 2>&1
  python -m tests.fuzz_post --first 3000 --count 200 > $OUT/fuzz_post.txt 2>&1
fi
# only what this script produced (other rounds' / tags' databases are read by tools/make_traffic_json.py, rocpd_pmc.py afterwards)
find $OUT gpurun_out/prof_r03 gpurun_out/sq_r03 -name '*.db' -size +20M -delete 2>/dev/null
ls -la $OUT
