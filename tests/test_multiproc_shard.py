"""CPU, world_size 2 over gloo: the N>1 path shards one picture's LCU rows over ranks with
no data-path collective; gathering the per-rank records must reproduce the unsharded result.
(On GPUs each rank runs svt_amd_me_picture_range_launch on its rows; here the oracle stands in
for the device so the sharding/gather logic is exercised without a GPU.)"""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import svtlib as S
from gpu_util import default_params


def lcu_rows_for_rank(h, rank, world):
    """Contiguous LCU-row bands, remainder rows to the first ranks (same rule bench.py uses)."""
    rows = (h + 63) // 64
    base, rem = divmod(rows, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _worker(rank, world, port, w, h, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    oracle = S.load_oracle()
    frames = [S.gen_luma("motion", w, h, t, 3) for t in range(3)]
    pics = [S.OraclePicture(oracle, f) for f in frames]
    p = default_params(w, h, num_lists=2, temporal_layer_index=1, cu8x8_mode=0)
    wl = (w + 63) // 64
    r0, r1 = lcu_rows_for_rank(h, rank, world)
    part = S.oracle_me_picture(oracle, p, pics[1], pics[0], pics[2], r0 * wl, r1 * wl)
    mine = torch.from_numpy(part[r0 * wl:r1 * wl].view(np.uint8).copy())
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine.numel()]))
    mx = int(max(s.item() for s in sizes))
    padded = torch.zeros(mx, dtype=torch.uint8)
    padded[:mine.numel()] = mine
    gathered = [torch.zeros(mx, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(gathered, padded)  # result hand-off only; the search itself needed no exchange
    if rank == 0:
        full = np.concatenate([g[:int(s.item())].numpy() for g, s in zip(gathered, sizes)])
        np.save(out_path, full)
    dist.barrier()
    dist.destroy_process_group()


def test_lcu_row_sharding_world2(tmp_path):
    w, h = 320, 328  # 5 x 6 LCUs, ragged bottom row -> 3 + 3 rows
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, w, h, out), nprocs=2, join=True)
    got = np.load(out).view(S.ME_LCU_DTYPE)
    oracle = S.load_oracle()
    frames = [S.gen_luma("motion", w, h, t, 3) for t in range(3)]
    pics = [S.OraclePicture(oracle, f) for f in frames]
    p = default_params(w, h, num_lists=2, temporal_layer_index=1, cu8x8_mode=0)
    want = S.oracle_me_picture(oracle, p, pics[1], pics[0], pics[2])
    assert got.shape == want.shape and got.tobytes() == want.tobytes()


def test_band_partition_covers_all_rows():
    for h in (64, 328, 1080, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            bands = [lcu_rows_for_rank(h, r, world) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == (h + 63) // 64
            assert all(bands[i][1] == bands[i + 1][0] for i in range(world - 1))
