# GPU box: SQ counters of the tracing kernel for one workload and kernel.  usage: collect_pmc.sh OUT.txt scene W H spp [kernel]
OUT=$1; SCENE=$2; W=$3; H=$4; SPP=$5; K=${6:-path_pool}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_sq1 /tmp/p_sq2
export DRT_KERNEL=$K
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d /tmp/p_sq1 -o sq -- python3 $R/tools/time_workload.py $SCENE $W $H $SPP > /tmp/tw1.txt 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d /tmp/p_sq2 -o sq -- python3 $R/tools/time_workload.py $SCENE $W $H $SPP > /tmp/tw2.txt 2>&1 &&
(grep -v amdgpu.ids /tmp/tw1.txt | tail -1; python3 $R/tools/pmc_summary.py /tmp/p_sq1 /tmp/p_sq2 | grep -v "resolve\|^ *$") > $OUT
