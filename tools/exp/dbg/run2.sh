set -x
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/t2 && sed "s#work_dirs/#/tmp/t2/#" configs/simple2_softmax_synthetic.yml > /tmp/t2/cfg.yml
EMBNET_DIST_BACKEND=gloo EMBNET_DUMP_FINAL_WEIGHTS=/tmp/t2/final_rank timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 tools/train.py /tmp/t2/cfg.yml --synthetic 10 --max_epochs 2 2>&1 | tail -40
