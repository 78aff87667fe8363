# Collect the round's judged evidence on one GPU box: the bench line (with cpu baseline), rocprofv3 kernel trace + stats of the
# SAME command line the driver uses (default --steps/--warmup), the PMC traffic passes and SQ / matrix-pipe counter passes.
# usage (from gpurun): bash tools/profile_round.sh r03
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}; O=$R/gpurun_out/profile_$TAG; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu > $O/bench_under_rocprof.json 2> $O/stats.err &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-also > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-also > $O/pmc_write.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1 -- python3 $R/bench.py --steps 5 --warmup 5 --no-cpu --no-also > $O/pmc_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq2 -- python3 $R/bench.py --steps 5 --warmup 5 --no-cpu --no-also > $O/pmc_sq2.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/set_fetch -- python3 $R/tools/run_path.py traffic_set 3 > $O/set_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/set_write -- python3 $R/tools/run_path.py traffic_set 3 > $O/set_write.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/set_sq -- python3 $R/tools/run_path.py traffic_set 3 > $O/set_sq.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- python3 $R/tools/run_path.py traffic_set 3 > $O/pmc_mfma.log 2>&1
echo exit=$?; cat $O/bench.json
