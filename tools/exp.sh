cd $GRAFT_REPO_ROOT
rm -f mpcholonavigation_amd/csrc/smpc_lane.o; make -C mpcholonavigation_amd/csrc > gpurun_out/mk.log 2>&1 || tail -5 gpurun_out/mk.log
for T in 64 60 32 16 4; do echo "== T=$T"; SMPC_PASS=lane timeout -k 10 200 python tools/ablate.py 65536 $T 2>&1 | grep "none\|all"; done
