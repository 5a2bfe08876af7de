set -e
export TMPDIR=/tmp
O=gpurun_out/ks_ab
rm -rf $O && mkdir -p $O
P="python3 tools/bench_conv.py --shapes v3_256_256 --ops fwd --reps 10"
for v in ks nks; do
  if [ $v = nks ]; then export SIV_LIB=tools/ab/libsiv_nks.so; fi
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${v}_fetch -- $P > $O/${v}_fetch.log 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/${v}_tcc -- $P > $O/${v}_tcc.log 2>&1
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --output-format csv -d $O/${v}_sq -- $P > $O/${v}_sq.log 2>&1
done
