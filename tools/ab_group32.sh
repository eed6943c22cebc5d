set -x
mkdir -p gpurun_out/r5d
for v in g16 g32 g16 g32; do
  FS_LIB_PATH=$PWD/libfastsparse_amd/build/variants/libfs_$v.so python bench.py --workload c2 --no-cpu-baseline --no-reproducible-cost > gpurun_out/r5d/c2_$v.$RANDOM.json 2> gpurun_out/r5d/c2_$v.err || echo FAIL $v
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5d/c2_g*.json')):
    r=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, r['roofline']['frac'], r['config']['A_mul_B_ms'], r['config']['At_mul_B_ms'], r['config']['self_check']['ok'], r['config'].get('one_time',{}).get('hbm_held_bytes'))
PY
