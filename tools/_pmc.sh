cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_m4/$name -o run -- python3 $R/tools/opbench.py --ops M4 --reps 2 > $R/gpurun_out/pmc_m4_$name.log 2>&1; }
run sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES &&
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES &&
run fetch FETCH_SIZE && run write WRITE_SIZE
ls $R/gpurun_out/pmc_m4/*
