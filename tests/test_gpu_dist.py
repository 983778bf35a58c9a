"""-m gpu: the RCCL path of tuatara_amd/dist.py in ONE process with an engine: torch's "nccl" backend is the RCCL bundled with the
torch ROCm-7.0 wheel, the engine links the system ROCm-7.2 runtime (check that the two coexist).  A GPU box has
one MI355X, so world_size = 1; world_size 2 runs over gloo in tests/test_dist_cpu.py and N = 2, 4, 8 in the driver's scaling bench
(python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...)."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nccl_all_gather_beside_an_engine(eng_bf16, funsd):
    import torch
    import torch.distributed as dist
    from tuatara_amd import dist as D
    from tuatara_amd.engine import DeviceBuffer
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    page = np.ascontiguousarray(funsd[:512, :384])
    before = eng_bf16.image_to_data(page)                       # engine first: HIP initialised by the ROCm 7.2 runtime
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        buf = DeviceBuffer(page.nbytes)
        buf.upload(page)
        res = eng_bf16.pages_to_data_dev(buf, 1, 512, 384)
        rec = D.pack_records(res)
        allrec = D.all_gather_records(rec, device="cuda")        # one RCCL collective, as in bench.py's step
        assert allrec.shape == (1,) + rec.shape and np.array_equal(allrec[0], rec)
        ids = D.recognise_sharded(np.zeros((5, 32, 128, 3), np.uint8), lambda c: eng_bf16.parseq_logits(c)[1], device="cuda")
        assert ids.shape == (5, 26)
        after = eng_bf16.image_to_data(page)                     # and the engine still works beside the process group
        assert [x["bbox"] for x in after] == [x["bbox"] for x in before] and [x["text"] for x in after] == [x["text"] for x in before]
        assert D.unpack_records(allrec[0])[0] == [x["ids"] for x in res[0]][:D.MAX_CROPS]
    finally:
        dist.destroy_process_group()
