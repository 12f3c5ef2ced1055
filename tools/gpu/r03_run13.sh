set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3m; mkdir -p $O
timeout 3000 python tests/t3_baseline_shapes.py --out $O/t3.json --fixture $O/t3_fixture.npz --digest $O/t3_digest.npz --ref-twice opt125m_w4g128,llama8b_w4g128 --cases opt125m_w4g128,llama8b_w4g128,llama8b_w4g128_full,llama8b_w2g32_asym_algext,llama8b_mxfp4,llama8b_nvfp4 > $O/t3.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3m/t3.json'))
for c in t['cases']:
    print(c['case'], c.get('error'), c.get('ref_wall_s'), c.get('ref_vs_ref'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused','alone_module','alone_fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff','hip_graph','targets_identical')})
PY
cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_opt -- python $GRAFT_REPO_ROOT/bench.py --workload opt-125m --no-cpu-baseline --no-extras --no-kernel-timing --steps 3 --warmup 1 > $GRAFT_REPO_ROOT/$O/opt_under_rocprof.json 2> $GRAFT_REPO_ROOT/$O/opt_under_rocprof.err; echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_opt -name "*.db" | head -1); echo $DB
python tools/rocprof_summary.py $DB --stats $O/opt125m_graph_kernel_stats.csv
head -25 $O/opt125m_graph_kernel_stats.csv | cut -c1-200
