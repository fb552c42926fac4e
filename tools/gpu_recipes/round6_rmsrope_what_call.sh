# round 6: WHAT are the wrong chunks the LDS-less narrow in-place RMSNorm + RoPE kernel (rrwg0 = rounds 3-5) leaves beside a neighbour's start-up / exit?
TAG=${TAG:-run}; ROUND=${ROUND:-r06}
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( for n in 1 2 3 4; do timeout 60 python tools/probes/dit_determinism.py CO 6 big > /dev/null 2>&1; done ) &
NB=$!
timeout 150 python tools/probes/rmsrope_twice.py V 60 --lib libwanhip_rrwg0.so --dump $O/rmsrope_wrong_rows 2>&1 | grep -E "launches differ|iteration|wrong chunks|Error|error" | cut -c1-700 | tee $O/${ROUND}_rmsrope_what_is_wrong_$TAG.log
wait $NB
