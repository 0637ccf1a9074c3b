set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06
for sk in 0 1 2 3; do
SMK_HALS_EP_SKIP=$sk timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/kt_ep$sk -o x -- python3 $R/bench.py --no-cpu-baseline --workload c3 --steps 20 --warmup 3 > /dev/null 2>&1
DB=$(find $OUT/kt_ep$sk -name '*.db' | head -1)
[ -n "$DB" ] && python3 $R/tools/prof_summary.py "$DB" $OUT/exp_ep_skip$sk.md > /dev/null
rm -rf $OUT/kt_ep$sk
echo "skip=$sk"; grep "hals_w_fused\|hals_sweep_pack" $OUT/exp_ep_skip$sk.md | cut -c1-30,90-160
done
cd $R
B="python3 bench.py --no-cpu-baseline"
for rep in 1 2; do for sh in 0 3; do
SMK_NNLS_G16_SHAPE=$sh $B --workload s_1m --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/exp_s1m_sh${sh}_r$rep.json
SMK_NNLS_G16_SHAPE=$sh $B --workload s_reuters --steps 200 --warmup 20 2>/dev/null | tail -1 > $OUT/exp_sr_sh${sh}_r$rep.json
SMK_NNLS_G16_SHAPE=$sh $B --workload b32 --steps 50 --warmup 5 2>/dev/null | tail -1 > $OUT/exp_b32_sh${sh}_r$rep.json
done; done
for f in $OUT/exp_*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f'%(j['value'],j['ms_per_step']))"; done
SMK_NNLS_G16_SHAPE=3 python3 tools/nnls_g16_check.py /tmp/x.npz | tail -1
