# the RCCL paths a one-GPU box can host: torch.distributed / library-owned communicator at world 1 -- all-gather and (round 4) all-to-all hooks
TAG=${TAG:-run}; ROUND=${ROUND:-r04}
cd $GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 180 python -c "import torch; print(torch.zeros(4).cuda().sum().item())" || { echo "GPU init failed"; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_sp.py -q -m gpu -p no:cacheprovider -k "world_1 or rccl" ) > $O/${ROUND}_pytest_rccl_world1_$TAG.log 2>&1; tail -5 $O/${ROUND}_pytest_rccl_world1_$TAG.log
