# does the update overlap pay?  same box: single-process graph / world-1 stage graphs with the updates on the side stream / on the main stream
for rep in 1 2; do
USE_GRAPH=1 python tools/ddp_world1_step.py 40 0 2>&1 | grep "ms/step"
USE_GRAPH=1 python tools/ddp_world1_step.py 40 1 2>&1 | grep "ms/step"
USE_GRAPH=1 SW_DDP_UPD_MAIN=1 python tools/ddp_world1_step.py 40 1 2>&1 | grep "ms/step"
done
