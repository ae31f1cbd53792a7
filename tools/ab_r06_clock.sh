cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do for v in _str5 _st; do echo "== stamps lib$v"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --stamps --variants 0 --shapes 128:128:64:8,256:128:64:8 2>&1 | grep stamps | cut -c1-1200; done; done
