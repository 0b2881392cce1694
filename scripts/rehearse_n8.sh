#!/bin/bash
# The driver's 8-GPU command rehearsed on a 1-GPU box: eight ranks at the STATED sizes (10 M frames per rank, the 100 M-frame
# configs[3] split) stacked on device 0.  Kernel times are 8x shared; what this measures is everything else of the N = 8 run --
# per-rank setup wall time and peak host memory, the rendezvous, the secondary blocks, the line -- and its total wall time.
#   gpurun -- 'bash scripts/rehearse_n8.sh [torchrun]'  ->  gpurun_out/r06_n8_stacked.json (+ _line.json, _wall.txt)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out; mkdir -p $O
free -g | head -2 > $O/r06_n8_wall.txt; nproc >> $O/r06_n8_wall.txt
ARGS="--gpus 8 --device-override 0 --steps 3 --warmup 1 --no-cpu-baseline --blocks-out $O/r06_n8_stacked.json"
if [ "$1" = "torchrun" ]; then
  CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py $ARGS"
else
  CMD="python bench.py $ARGS"
fi
echo "$CMD" >> $O/r06_n8_wall.txt
SECONDS=0
timeout ${REHEARSE_TIMEOUT:-1500} $CMD > $O/r06_n8_line.json 2> $O/r06_n8_err.txt < /dev/null
echo "rc $? wall_s $SECONDS" >> $O/r06_n8_wall.txt
cat $O/r06_n8_wall.txt
tail -c 1600 $O/r06_n8_line.json
grep -v '^{' $O/r06_n8_err.txt | tail -5 | cut -c1-300
