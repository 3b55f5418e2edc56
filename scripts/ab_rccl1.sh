#!/bin/bash
# Same-box A/B of the one-rank RCCL rehearsal (PCGAN_FORCE_COLLECTIVES=1: RCCL refuses two ranks on one device) against the plain
# step: what the collective side of the step costs before any fabric is involved.  usage: bash scripts/ab_rccl1.sh [rounds]
set -o pipefail
R=${1:-2}
mkdir -p gpurun_out
OUT=gpurun_out/ab_rccl1.txt
: > $OUT
run() {   # label, env...
    local label=$1; shift
    env "$@" python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-experiment > gpurun_out/_ab.json 2> gpurun_out/_ab.err || { tail -5 gpurun_out/_ab.err; exit 1; }
    python - "$label" <<'PY' >> gpurun_out/ab_rccl1.txt
import json, sys
lines = [l for l in open('gpurun_out/_ab.json') if l.strip()]
assert len(lines) == 1, 'stdout must carry the one JSON line: %r' % lines
d = json.loads(lines[0])
pr = (d.get('per_rank') or [{}])[0]
print('%-34s %8.1f img/s %7.2f ms/step  host %5.2f ms  allreduce %s ms/step (%s per step)' % (
    sys.argv[1], d['value'], d['ms_per_step'], d.get('host_issue_ms_per_step', -1), pr.get('allreduce_ms_per_step'), pr.get('allreduces_per_step')))
PY
}
for r in $(seq $R); do
    run "plain" PCGAN_FORCE_COLLECTIVES=0
    run "rccl world=1 sequential" PCGAN_FORCE_COLLECTIVES=1
    run "rccl world=1 overlapped" PCGAN_FORCE_COLLECTIVES=1 PCGAN_DDP_OVERLAP=1
    run "rccl world=1 on gradient stream" PCGAN_FORCE_COLLECTIVES=1 PCGAN_DDP_GRAD_STREAM=1
    run "rccl world=1 seq + hash check/5" PCGAN_FORCE_COLLECTIVES=1 PCGAN_DDP_CHECK=5
done
cat $OUT
