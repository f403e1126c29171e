cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for a in 0 7; do
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/clk_$a -- ./tools/conv_bench_abl$a 2 8 512 512 -1 -1 20 > gpurun_out/clk_$a.log 2>&1
done
