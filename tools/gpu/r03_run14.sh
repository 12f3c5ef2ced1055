set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3n; mkdir -p $O
python tools/gpu/r03_exp_det.py plain > $O/det_plain.json 2>/dev/null; cat $O/det_plain.json
AR_DW_VIA_TEMP=1 python tools/gpu/r03_exp_det.py temp > $O/det_temp.json 2>/dev/null; cat $O/det_temp.json
python tools/gpu/r03_exp_det.py mfma > $O/det_mfma.json 2>/dev/null; cat $O/det_mfma.json
