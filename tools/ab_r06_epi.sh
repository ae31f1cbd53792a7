# round 6: same-box A/B of the Winograd conv epilogue (lib = the working tree; AB_LIBS e.g. "_r5" = round-5 kernel) + stamps (_st)
cd ${GRAFT_REPO_ROOT:-.}
python -m pytest tests/test_gpu_wino.py -x -q -m gpu 2>&1 | tail -3
SH=${SH:-128:128:64:8,256:128:64:8,256:256:32:8}
for rep in 1 2; do for v in "" $AB_LIBS; do echo "== lib$v (res)"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --variants 0 --shapes $SH 2>&1 | grep "A/B" | cut -c1-175; done; done
for v in "" $AB_LIBS; do echo "== lib$v (no-res)"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --no-res --variants 0 --shapes $SH 2>&1 | grep "A/B" | cut -c1-175; done
for v in $ST_LIBS; do echo "== stamps lib$v"; MD_LIB=$PWD/meshdiffusion_amd/libmeshdiffusion_hip$v.so python tools/bench_wino.py --f8 --stamps --variants 0 --shapes 128:128:64:8,256:128:64:8 2>&1 | grep stamps | cut -c1-900; done
