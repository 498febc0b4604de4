#!/bin/bash
# Evidence of one round, run on the GPU box through gpurun:  bash tools/collect_evidence.sh r03_a
# -> gpurun_out/<tag>/: bench_default.json, kernel stats of a 35-step profiled run, PMC passes (FETCH_SIZE / WRITE_SIZE / SQ
# counters, each in its own run with --kernel-trace only), the library calibration table.
TAG=${1:-r05}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /root/repo
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 300 python tools/blas_ref.py > $OUT/blas_ref.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-extra-step --no-per-shape --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof -o c2a -- $B --steps 35 --warmup 5 > $OUT/bench_profiled_run.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/traffic/f -o p --output-format csv -- $B --steps 9 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/traffic/w -o p --output-format csv -- $B --steps 9 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -d $OUT/sq/a -o p --output-format csv -- $B --steps 9 --warmup 2 > /dev/null 2>&1
cd /root/repo
python tools/rocpd_stats.py $OUT/prof/c2a_results.db > $OUT/kernel_stats.csv 2> $OUT/kernel_stats.err
python tools/rocpd_timeline.py $OUT/prof/c2a_results.db 0 -2 > $OUT/timeline_c2a.txt 2>&1
python tools/pmc_traffic.py $OUT/traffic $OUT/pmc_traffic.json > /dev/null 2> $OUT/pmc_traffic.err
python tools/pmc_summary.py $OUT/sq gemm_nt_p8 gemm_nt_ldp gemm_nt_ld gemm_nt_pp wgrad_p8 wgrad_ld attn_fwd attn_bwd_rows attn_keep_bits ln_bwd ln_fwd8 gemm_nt_h16 > $OUT/pmc_sq_summary.txt 2>&1
(hostname; cat /proc/loadavg; nproc) > $OUT/box.txt 2>&1
python tools/evidence_summary.py $OUT 0 $OUT/summary.json > $OUT/summary.out 2>&1
rm -rf $OUT/prof $OUT/traffic $OUT/sq
ls -la $OUT
# secondary configurations: kernel stats of the C4 training step and of C1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c4 -o c4 -- python /root/repo/tools/prof_secondary.py C4 16 30 --det-train > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_c1 -o c1 -- python /root/repo/tools/prof_secondary.py C1 64 30 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_b8 -o b8 -- python /root/repo/tools/prof_secondary.py C2a 8 40 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_tr -o tr -- python /root/repo/tools/prof_secondary.py C2a 64 40 --rec-train > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $OUT/sq_c4/a -o p --output-format csv -- python /root/repo/tools/prof_secondary.py C4 16 8 --det-train > /dev/null 2>&1
cd /root/repo
python tools/rocpd_stats.py $(find $OUT/prof_c4 -name "*.db" | head -1) > $OUT/c4_train_kernel_stats.csv 2> $OUT/c4.err
python tools/rocpd_stats.py $(find $OUT/prof_c1 -name "*.db" | head -1) > $OUT/c1_kernel_stats.csv 2> $OUT/c1.err
python tools/rocpd_stats.py $(find $OUT/prof_b8 -name "*.db" | head -1) > $OUT/c2a_b8_kernel_stats.csv 2> $OUT/b8.err
python tools/rocpd_stats.py $(find $OUT/prof_tr -name "*.db" | head -1) > $OUT/c2a_train_kernel_stats.csv 2> $OUT/tr.err
python tools/rocpd_timeline.py $(find $OUT/prof_b8 -name "*.db" | head -1) 0 -2 > $OUT/timeline_c2a_b8.txt 2>&1
python tools/pmc_summary.py $OUT/sq_c4 gemm_nt_ldp gemm_nt_ld wgrad_ld attn_fwd attn_bwd_rows attn_bwd_keys ln_bwd ln_fwd8 focal assemble_bwd > $OUT/c4_pmc_sq_summary.txt 2>&1
rm -rf $OUT/prof_c4 $OUT/prof_c1 $OUT/sq_c4 $OUT/prof_b8 $OUT/prof_tr
ls -la $OUT
