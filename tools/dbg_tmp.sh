cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/r03s_dbg.log 2>&1
import re,sys
src=open('tests/test_sharded.py').read()
m=re.search(r"NCCL_WORKER = r'''(.*?)'''", src, re.S)
open('/tmp/nccl_worker.py','w').write(m.group(1))
PY
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29653 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
echo "== default" >> gpurun_out/r03s_dbg.log
PFP_VERBOSE=1 timeout -k 10 300 python /tmp/nccl_worker.py $GRAFT_REPO_ROOT >> gpurun_out/r03s_dbg.log 2>&1
echo "== group rows 0" >> gpurun_out/r03s_dbg.log
PFP_TEST_HOOKS=1 PFP_EMIT_GROUP_ROWS=0 timeout -k 10 300 python /tmp/nccl_worker.py $GRAFT_REPO_ROOT >> gpurun_out/r03s_dbg.log 2>&1
echo "== done" >> gpurun_out/r03s_dbg.log
