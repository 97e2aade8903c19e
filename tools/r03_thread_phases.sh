# where k_poa_thread's time goes: lane-0 cycle counters per phase (CCSX_PROFILE_PHASES build; timing-only, the product build has no timers)
cd $GRAFT_REPO_ROOT
CCSX_EXTRA_FLAGS="-DCCSX_PROFILE_PHASES=1" python -c "import __graft_entry__ as g; g.build(force=True)" > /tmp/build.log 2>&1 || tail -5 /tmp/build.log
timeout 300 python - <<'PY' 2>&1 | grep "thr \|done"
from ccs_amd import api
b = api.synth(2048, 10, 10000, seed=5)
h = api.Handle(0)
r = h.consensus(b)
r = h.consensus(b)
print("done", int((r.status == 0).sum()))
h.close()
PY
python -c "import __graft_entry__ as g; g.build(force=True)" > /dev/null 2>&1
