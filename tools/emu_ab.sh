cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/h12
for rep in 1 2; do
for case in "head|TH_X=0" "nopregrid|TH_PREGRID=0" "la2|TH_LOOKAHEAD=2" "hullseq|TH_HULL_SEQ=1"; do
  name=${case%%|*}; envs=${case#*|}
  env $envs python bench.py --emulate-world 8 --emulate-rank 0 --no-cpu-baseline --no-extras > gpurun_out/h12/$name$rep.json 2> gpurun_out/h12/$name$rep.err
  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().split(chr(10))[-1]); print(sys.argv[1], round(d['ms_per_step'],3), d.get('host_queue_ms_per_step'), d.get('host_wait_ms_per_step'))" gpurun_out/h12/$name$rep.json
done; done
