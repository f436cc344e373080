set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof
R=$PWD
mkdir -p $O
C1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cd /tmp
for s in conv4_2 conv3_2 m384_128 m128 m288_96 m96 m6_512 m7_52 pair512 stem; do
  timeout -s KILL 120 rocprofv3 --pmc $C1 --output-format csv -d $O/pmc_conv_$s -- python3 $R/tools/conv_bench.py --reps 2 --shapes $s > $O/pmc_conv_$s.log 2>&1
done
cd $R
: > $O/pmc_conv.md
for s in conv4_2 conv3_2 m384_128 m128 m288_96 m96 m6_512 m7_52 pair512 stem; do
  echo "shape $s" >> $O/pmc_conv.md
  python3 tools/pmc_table.py $O/pmc_conv_$s 'conv3x3|conv_stem|conv1x1_pair' >> $O/pmc_conv.md
done
find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/pmc_conv.md
