set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
A="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
B="SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --pmc $A -d $OUT/a -o a -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-timed-s 0 > /dev/null 2> $OUT/a.err
rocprofv3 --kernel-trace --pmc $B -d $OUT/b -o b -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --min-timed-s 0 > /dev/null 2> $OUT/b.err
cd $ROOT
python tools/pmc_kernel.py $(find $OUT/a -name "*.db" | head -1) $A > $OUT/a.txt 2>&1
python tools/pmc_kernel.py $(find $OUT/b -name "*.db" | head -1) $B > $OUT/b.txt 2>&1
rm -rf $OUT/a $OUT/b
cat $OUT/a.txt $OUT/b.txt
