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


class _DeviceView:
    """Foreign device memory (a pointer handed out by the C-ABI) as something torch can alias without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False),
                                         "strides": None, "version": 2}


def device_bytes(torch, ptr, nbytes, device="cuda"):
    """uint8 tensor ALIASING [ptr, ptr + nbytes) of device memory owned by the library."""
    return torch.as_tensor(_DeviceView(ptr, nbytes), device=device)


def broadcast_inplace(buf, world, dist, src=0):
    """One broadcast of `buf` (a tensor every rank holds with the same size): rank `src`'s contents land in the
    other ranks' `buf`, device to device when the tensors are device memory (RCCL over xGMI)."""
    if world > 1:
        dist.broadcast(buf, src)
    return buf


KINDS = (("phone", 1, "ReadPhoneExtractorParameters", "phone_extractor.bin"),
         ("pitch", 2, "ReadPitchEstimatorParameters", "pitch_estimator.bin"),
         ("wave", 3, "ReadWaveformGeneratorParameters", "waveform_generator.bin"),
         ("embed", 4, "ReadEmbeddingSetterParameters", "embedding_setter.bin"))


def all_ranks_ok(ok, world, dist, torch, device):
    """True only if `ok` is true on every rank (MIN all-reduce): ranks must take the same branch before a collective."""
    if world == 1:
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


LOADERS = {"phone": "BeatriceHip_LoadPhoneExtractorFromMemory", "pitch": "BeatriceHip_LoadPitchEstimatorFromMemory",
           "wave": "BeatriceHip_LoadWaveformGeneratorFromMemory", "embed": "BeatriceHip_LoadEmbeddingSetterFromMemory"}


def load_models_from_rank0(product, objs, model_dir, rank, world, dist, torch, device="cuda"):
    """The four model objects of this rank (objs: name -> handle, freshly created).  Rank 0 reads and packs the files
    (Read*Parameters); every other rank gets an empty device blob of the same size (BeatriceHip_ModelBlob) that
    receives rank 0's PACKED blob by one broadcast each, device to device -- no file, no host copy, no repacking --
    and is then marked ready.  If any rank cannot expose its blobs to torch, ALL ranks agree (all_ranks_ok) to ship
    the file bytes instead (broadcast_bytes + BeatriceHip_Load*FromMemory).  Returns (bytes broadcast, path taken)."""
    import ctypes as C
    views, ok = {}, True
    for name, kind, reader, fname in KINDS:
        if rank == 0:
            err = getattr(product, reader)(objs[name], os.path.join(model_dir, fname).encode())
            if err:
                raise RuntimeError("%s: Beatrice_ErrorCode %d" % (fname, err))
        if world == 1:
            continue
        try:
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            rc = product.BeatriceHip_ModelBlob(kind, objs[name], 0 if rank == 0 else 1, C.byref(ptr), C.byref(nbytes))
            if rc:
                raise RuntimeError("BeatriceHip_ModelBlob(%s): %d" % (name, rc))
            views[name] = device_bytes(torch, ptr.value, nbytes.value, device)
        except Exception as e:  # noqa: BLE001 -- any failure means "take the other path", decided collectively below
            print("shard: rank %d cannot share the %s blob in place (%s)" % (rank, name, e))
            ok = False
    if world == 1:
        return 0, "file"
    if all_ranks_ok(ok, world, dist, torch, device):
        moved = 0
        for name, kind, _, _ in KINDS:
            broadcast_inplace(views[name], world, dist)
            moved += views[name].numel()
            if rank != 0:
                torch.cuda.synchronize()
                rc = product.BeatriceHip_ModelBlobReady(kind, objs[name])
                if rc:
                    raise RuntimeError("BeatriceHip_ModelBlobReady(%s): %d" % (name, rc))
        return moved, "device blobs, in place"
    moved = 0
    for name, kind, _, fname in KINDS:
        data = open(os.path.join(model_dir, fname), "rb").read() if rank == 0 else b""
        data = broadcast_bytes(data, rank, world, dist, torch, device)
        moved += len(data)
        if rank != 0:
            fn = getattr(product.lib, LOADERS[name])
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
            err = fn(objs[name], data, len(data))
            if err:
                raise RuntimeError("%s: Beatrice_ErrorCode %d" % (LOADERS[name], err))
    return moved, "file bytes, repacked per rank"


def share_speaker_tables(product, batch_handle, n_speakers, rank, world, dist, torch, device="cuda", host_tables=None):
    """Rank 0 has uploaded its tables (BeatriceBatch_SetSpeakerTables); the raw device tables go to the other ranks
    by four broadcasts and are projected there (BeatriceBatch_ProjectSpeakerTables).  Fallback, agreed by all ranks:
    rank 0's host arrays (host_tables: the SpeakerTables object, rank 0 only) are broadcast and every other rank calls
    BeatriceBatch_SetSpeakerTables itself.  Returns (bytes broadcast, path taken)."""
    import ctypes as C
    import numpy as np
    if world == 1:
        return 0, "file"
    views, ok = [], True
    try:
        ptrs, sizes = (C.c_void_p * 4)(), (C.c_size_t * 4)()
        rc = product.BeatriceBatch_SpeakerTablesDevice(batch_handle, ptrs, sizes)
        if rc:
            raise RuntimeError("BeatriceBatch_SpeakerTablesDevice: %d" % rc)
        views = [device_bytes(torch, ptrs[i], sizes[i], device) for i in range(4)]
    except Exception as e:  # noqa: BLE001
        print("shard: rank %d cannot share its speaker tables in place (%s)" % (rank, e))
        ok = False
    if all_ranks_ok(ok, world, dist, torch, device):
        for v in views:
            broadcast_inplace(v, world, dist)
        if rank != 0:
            torch.cuda.synchronize()
            rc = product.BeatriceBatch_ProjectSpeakerTables(batch_handle, n_speakers)
            if rc:
                raise RuntimeError("BeatriceBatch_ProjectSpeakerTables: %d" % rc)
        return int(sum(v.numel() for v in views)), "device tables, in place"
    shapes = ((n_speakers, 512, 128), (n_speakers, 256), (9, 256), (n_speakers, 384, 128))  # codebooks, additive, formant, kv
    arrays, moved = [], 0
    for i, shape in enumerate(shapes):
        src = (host_tables.codebooks, host_tables.additive, host_tables.formant, host_tables.kv)[i] if rank == 0 else None
        t = torch.from_numpy(np.ascontiguousarray(src, dtype=np.float32)).to(device) if rank == 0 else \
            torch.empty(shape, dtype=torch.float32, device=device)
        dist.broadcast(t, 0)
        arrays.append(np.ascontiguousarray(t.cpu().numpy()))
        moved += t.numel() * 4
    if rank != 0:
        fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))  # noqa: E731
        rc = product.BeatriceBatch_SetSpeakerTables(batch_handle, n_speakers, fp(arrays[0]), fp(arrays[1]), fp(arrays[2]), fp(arrays[3]))
        if rc:
            raise RuntimeError("BeatriceBatch_SetSpeakerTables: %d" % rc)
    return moved, "host tables, projected per rank"


def affine_speaker(rank, world, local_stream, n_speakers):
    """Speaker-affine placement: rank r starts its streams on the speakers congruent to r
    modulo the world size, so a GPU keeps 1/world of the codebooks and K/V tables hot.  With fewer speakers than
    ranks every rank cycles through all of them."""
    if n_speakers < world:
        return local_stream % n_speakers
    # the speakers congruent to `rank`: rank, rank + world, ... below n_speakers -- ceil((n_speakers - rank) / world) of
    # them, so that the ranks' sets PARTITION the table for any n_speakers (64 over 3 ranks: 22 + 21 + 21)
    mine = (n_speakers - rank + world - 1) // world
    return rank + world * (local_stream % mine)


def max_over_ranks(value, world, dist, torch, device):
    if world == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_over_ranks(value, world, dist, torch, device):
    """every rank's value, in rank order (all ranks receive the list)"""
    if world == 1:
        return [value]
    t = torch.tensor([value], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(x.item()) for x in out]
