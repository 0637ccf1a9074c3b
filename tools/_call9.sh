set -u
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r06; mkdir -p $OUT
B="python3 bench.py --no-cpu-baseline"
for rep in 1 2 3; do for g in 0 1; do
SMK_NNLS_G16=$g $B --workload c4s --steps 20 --warmup 3 2>/dev/null | tail -1 > $OUT/g64_c4s_g${g}_r$rep.json
done; done
for rep in 1 2; do for g in 0 1; do
SMK_NNLS_G16=$g $B --emulate-world 8 2>/dev/null | tail -1 > $OUT/g64_emu8_g${g}_r$rep.json
SMK_NNLS_G16=$g $B --emulate-world 8 --data planted 2>/dev/null | tail -1 > $OUT/g64_emu8p_g${g}_r$rep.json
done; done
for g in 0 1; do
SMK_NNLS_G16=$g python3 tools/active_pivoting.py 262144 65536 64 25 planted 8 2>&1 | grep -v "^\[" | head -2 | cut -c1-400 > $OUT/g64_piv8_g$g.txt
SMK_NNLS_G16=$g python3 tools/active_pivoting.py 16384 8192 64 30 both 2>&1 | grep -v "^\[" | grep -v "ms per" > $OUT/g64_mid_g$g.txt
SMK_NNLS_G16=$g python3 tools/active_pivoting.py 16384 8192 48 30 both 2>&1 | grep -v "^\[" | grep -v "ms per" >> $OUT/g64_mid_g$g.txt
done
for f in $OUT/g64_*.json; do echo $f; python3 -c "
import json
j=json.loads(open('$f').read()); print('  it/s %.2f ms/step %.4f'%(j['value'],j['ms_per_step']))"; done
cat $OUT/g64_piv8_g0.txt $OUT/g64_piv8_g1.txt $OUT/g64_mid_g0.txt $OUT/g64_mid_g1.txt
