mkdir -p gpurun_out
timeout 600 python tools/dbg/r4_tokfwd_probe.py 2>&1 | grep -v Warn | tail -5
bash tools/prof_step.sh r4g_fsq > gpurun_out/prof21.log 2>&1
sed -n 5,9p gpurun_out/r4g_fsq_step_seq.txt | cut -c1-100
head -1 gpurun_out/r4g_fsq_step_summary.txt
