#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_cg2; rm -rf $O; mkdir -p $O
timeout -k 10 150 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/p4 -- python3 profiles/tools/prof_cgb.py 8 0 > $O/p4.log 2>&1; echo p4 rc=$? | tee -a $O/status.log
timeout -k 10 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $O/p5 -- python3 profiles/tools/prof_cgb.py 8 0 > $O/p5.log 2>&1; echo p5 rc=$? | tee -a $O/status.log
timeout -k 10 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d $O/p6 -- python3 profiles/tools/prof_cgb.py 8 0 > $O/p6.log 2>&1; echo p6 rc=$? | tee -a $O/status.log
