set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3h; mkdir -p $O
python tests/asym_grad_probe.py 2 32 > $O/asym_w2g32.json 2> $O/asym.err; python -c "
import json; d=json.load(open('$O/asym_w2g32.json'))
print(d['kernel_vs_autograd'])
for e in d['dV_examples']: print(e)
for e in d['dmax_examples']: print(e)
"; tail -2 $O/asym.err
