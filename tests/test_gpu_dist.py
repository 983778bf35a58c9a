"""-m gpu: the RCCL path of tuatara_amd/dist.py in ONE process with an engine: torch's "nccl" backend is the RCCL bundled with the
torch ROCm-7.0 wheel, the engine links the system ROCm-7.2 runtime (check that the two coexist).  They do when
torch's runtime comes up first (bench.py's order; Engine() sees to it when torch is already imported).  A GPU box has one MI355X,
so world_size = 1; world_size 2 runs over gloo in tests/test_dist_cpu.py and N = 2, 4, 8 in the driver's scaling bench
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...).
Each case runs in a fresh interpreter so that the initialisation order is the one written here."""
import os
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu

BODY = r'''
import os, socket, sys
import numpy as np
sys.path.insert(0, {root!r})
{first}
from tuatara_amd import dist as D
from tuatara_amd.engine import DeviceBuffer, Engine
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
eng = Engine({wdir!r}, precision="bf16")
from PIL import Image
page = np.ascontiguousarray(np.array(Image.open({png!r}).convert("RGB"))[:512, :384])
before = eng.image_to_data(page)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
buf = DeviceBuffer(page.nbytes); buf.upload(page)
res = eng.pages_to_data_dev(buf, 1, 512, 384)
rec = D.pack_records(res)
allrec = D.all_gather_records(rec, device="cuda")            # one RCCL collective, as in bench.py's step
assert allrec.shape == (1,) + rec.shape and np.array_equal(allrec[0], rec)
ids = D.recognise_sharded(np.zeros((5, 32, 128, 3), np.uint8), lambda c: eng.parseq_logits(c)[1], device="cuda")
assert ids.shape == (5, 26)
after = eng.image_to_data(page)                               # the engine still works beside the process group
assert [x["bbox"] for x in after] == [x["bbox"] for x in before] and [x["text"] for x in after] == [x["text"] for x in before]
assert D.unpack_records(allrec[0])[0] == [x["ids"] for x in res[0]][:D.MAX_CROPS] and len(before) > 3
dist.destroy_process_group()
print("OK", len(before))
'''


@pytest.mark.parametrize("first", ["import torch; torch.cuda.set_device(0)      # bench.py's order: torch's runtime first",
                                   "import torch                                  # imported only: Engine() initialises torch's runtime before its own"])
def test_nccl_all_gather_beside_an_engine(weights, first):
    code = BODY.format(root=ROOT, wdir=weights["dir"], png=os.path.join(ROOT, "tests", "data", "funsd_0001129658.png"), first=first)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
