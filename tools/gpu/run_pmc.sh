# rocprofv3 counter passes for profiles/r03 (each --pmc pass is its own process; kernel-trace separately)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
step() { log=$1; shift; "$@" > $O/$log 2>&1; rc=$?; echo "[$log] rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout/kill: stopping"; tail -n 5 $O/$log; exit $rc; fi; }
BENCH1="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-kernel-roofline --no-graph --no-config2 --no-fp8-point --no-driver-points"
step b1_fetch.log timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE -M --output-format csv -d $O/b1_fetch -- $BENCH1
step b1_write.log timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE -M --output-format csv -d $O/b1_write -- $BENCH1
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"
for t in 88 98; do
step conv_t${t}_sq.log timeout -k 10 300 rocprofv3 --pmc $SQ -M --output-format csv -d $O/conv_t${t}_sq -- python $R/tools/profile_gemm.py 128 10 $t
step conv_t${t}_fetch.log timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -M --output-format csv -d $O/conv_t${t}_fetch -- python $R/tools/profile_gemm.py 128 10 $t
step conv_t${t}_write.log timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -M --output-format csv -d $O/conv_t${t}_write -- python $R/tools/profile_gemm.py 128 10 $t
step conv_t${t}_trace.log timeout -k 10 300 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/conv_t${t}_trace -- python $R/tools/profile_gemm.py 128 10 $t
done
SQA="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU"
step attn_sq.log timeout -k 10 300 rocprofv3 --pmc $SQA -M --output-format csv -d $O/attn_sq -- python $R/tools/profile_attn.py 128
step attn_trace.log timeout -k 10 300 rocprofv3 --kernel-trace --stats -M --output-format csv -d $O/attn_trace -- python $R/tools/profile_attn.py 128
cd $R
for d in b1_fetch b1_write conv_t88_sq conv_t88_fetch conv_t88_write conv_t98_sq conv_t98_fetch conv_t98_write attn_sq; do
  f=$(find $O/$d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f --json $O/$d.json > $O/$d.txt 2>&1
done
find $O -name "*kernel_stats.csv" | head; ls $O/*.txt
# drop the big raw CSVs (the merge back is limited to 64 MiB)
find $O -name "*counter_collection.csv" -size +20M -delete
find $O -name "*kernel_trace.csv" -delete
