# usage: pmc2.sh <path for run_path.py> <outdir-under-gpurun_out> [kernel filter]  -- kernel stats + SQ / matrix-pipe / LDS counter passes
R=$GRAFT_REPO_ROOT; P=$1; O=$R/gpurun_out/$2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/run_path.py $P 3 > $O/stats.log 2>&1 &&
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc1 -- python3 $R/tools/run_path.py $P 2 > $O/pmc1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU --output-format csv -d $O/pmc2 -- python3 $R/tools/run_path.py $P 2 > $O/pmc2.log 2>&1 &&
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc3 -- python3 $R/tools/run_path.py $P 2 > $O/pmc3.log 2>&1
echo exit=$?
python3 $R/tools/pmc_summary.py $O "$3" > $O/summary.txt 2>&1; cat $O/summary.txt
