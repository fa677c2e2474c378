"""Worker of test_parity_gpu.py::test_sharded_march_single_rank_nccl (needs a GPU)."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

from heat_amd import modeldict as mdl
from heat_amd.sharded import ShardedMarch
from oracle import oracle as orc


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    md, st = mdl.ragged_mixed(800, Z=8, dt=45.0, seed=21)
    w = mdl.weather_series(8, 45.0)
    ref = st.copy()
    rc, _ = orc.OracleModel(md).march(ref, w)
    assert rc == 0
    sm = ShardedMarch(md, 0, 1, device_index=0)
    got = st.copy()
    sm.batch.upload_state(got)
    sm.march_resident(w[:3])
    sm.march_resident(w[3:])
    sm.synchronize()
    sm.batch.download_state(got)
    sm.close()
    dist.destroy_process_group()
    for idx in (mdl.node_slots(md), md["hs_front_slot"], md["hs_back_slot"], md["flow_front_slot"],
                md["flow_back_slot"], md["zone_slot"]):
        assert np.allclose(got[idx], ref[idx], rtol=1e-9, atol=1e-9)
    print("SHARDED OK")


if __name__ == "__main__":
    main()
