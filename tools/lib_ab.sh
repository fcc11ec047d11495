# tools/lib_ab.sh: same-box A/B of two builds of the library (copy them to transhuman_amd/lib_before.so.bin / lib_after.so.bin first):
# frame ms, render_fast ms, cycles per tile and clock inside the fused kernel, interleaved three times
cd $GRAFT_REPO_ROOT
run() { timeout 200 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=d.get('fused_kernel_inside'); print(round(d['ms_per_step'],3), round(d.get('dropin_ms_per_step',0),3), round(f['cycles_per_tile']), round(f['clock_GHz_inside_the_launch'],3))"; }
for rep in 1 2 3; do
  cp transhuman_amd/lib_before.so.bin transhuman_amd/libtranshuman_hip.so; echo "before $(run)"
  cp transhuman_amd/lib_after.so.bin transhuman_amd/libtranshuman_hip.so; echo "after  $(run)"
done
