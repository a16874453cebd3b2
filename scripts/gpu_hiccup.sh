cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-frontend > gpurun_out/h.json 2> gpurun_out/h.err
  python3 -c "import json; d=json.loads(open('gpurun_out/h.json').read().strip().splitlines()[-1]); print('short', d['ms_per_step'], d['stage_ms']['screens']+d['stage_ms']['tod_synthesis_pipelined'])"
done
for i in 1 2; do
  timeout -k 10 500 python3 bench.py > gpurun_out/h.json 2> gpurun_out/h.err
  python3 -c "import json; d=json.loads(open('gpurun_out/h.json').read().strip().splitlines()[-1]); print('full', d['ms_per_step'], d['stage_ms']['screens']+d['stage_ms']['tod_synthesis_pipelined'])"
done
