"""Multi-process plumbing for the data-parallel path: one process per GPU,
streams sharded by rank, model bytes broadcast once from rank 0 (RCCL over xGMI when the backend is
nccl; gloo on CPU in the tests), no collective inside a hop."""
import os

MODEL_FILES = ("phone_extractor.bin", "pitch_estimator.bin", "waveform_generator.bin",
               "embedding_setter.bin", "speaker_embeddings.bin")


def stream_range(rank, world, total_streams):
    """Contiguous shard [lo, hi) of `total_streams` for `rank`; the remainder goes to the first ranks."""
    base, rem = divmod(total_streams, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_bytes(data, rank, world, dist, torch, device):
    """rank 0's `data` (bytes) -> every rank, as ONE broadcast of a uint8 tensor on `device`."""
    if world == 1:
        return data
    n = torch.tensor([len(data) if rank == 0 else 0], dtype=torch.int64, device=device)
    dist.broadcast(n, 0)
    if rank == 0:
        buf = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=device)
    dist.broadcast(buf, 0)
    return buf.cpu().numpy().tobytes()


def broadcast_model(model_dir, rank, world, dist, torch, device):
    """Returns {file name: bytes} on every rank; only rank 0 needs `model_dir` populated."""
    blobs = {}
    for f in MODEL_FILES:
        data = open(os.path.join(model_dir, f), "rb").read() if rank == 0 else b""
        blobs[f] = broadcast_bytes(data, rank, world, dist, torch, device)
    return blobs


def max_over_ranks(value, world, dist, torch, device):
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
