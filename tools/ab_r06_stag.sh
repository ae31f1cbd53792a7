# round 6: are the CUs in lockstep (raw stamps dumped for the phase histogram), and does a first-generation stagger help?
cd ${GRAFT_REPO_ROOT:-.}
SH=128:128:64:8,256:128:64:8,256:256:32:8
for rep in 1 2; do for v in "" _sg1 _sg2; do echo "== lib$v (res)"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --variants 0 --shapes $SH 2>&1 | grep "A/B" | cut -c1-175; done; done
for v in _st _stsg1; do echo "== stamps lib$v"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --stamps --dump gpurun_out/stamps$v --variants 0 --shapes 128:128:64:8 2>&1 | grep stamps | cut -c1-900; done
