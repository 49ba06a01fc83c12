"""Multi-GPU sharding of independent video streams (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on ROCm, "gloo" in the CPU tests).  The reference is single-process / single-GPU
(test_online_tra.py:160-161); independent video pairs have no data dependency, so streams are dealt round-robin to
ranks and the only collective is one all_gather of a small per-rank record at the end of a run."""
import torch


def shard_streams(n_streams, rank, world_size):
    """Stream ids handled by `rank`: i with i % world_size == rank (stream i -> GPU i mod N)."""
    if not 0 <= rank < world_size:
        raise ValueError('rank %d outside world of %d' % (rank, world_size))
    return list(range(rank, n_streams, world_size))


def gather_records(record, dist=None, device=None, force_collective=False):
    """record: 1-D float64 tensor of fixed length -> [world, len] tensor on the host (identity without dist).
    A world of one rank skips the collective unless `force_collective` (the one-rank RCCL smoke test: init `nccl` with
    world size 1 and push the record through a real device-side all_gather)."""
    record = record.to(torch.float64)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not force_collective):
        return record.detach().cpu().unsqueeze(0)
    if device is not None:
        record = record.to(device)
    out = [torch.zeros_like(record) for _ in range(dist.get_world_size())]
    dist.all_gather(out, record)
    return torch.stack(out).cpu()


def collective_backend_version():
    """'rccl x.y.z' of the library behind torch.distributed's "nccl" backend on ROCm (None when unavailable)."""
    try:
        v = torch.cuda.nccl.version()
        return 'rccl ' + '.'.join(str(x) for x in v)
    except Exception:
        return None


def aggregate_fps(records):
    """records [world, >=2] with columns (frames, seconds, ...): whole-job fps = sum(frames) / max(seconds)."""
    return float(records[:, 0].sum()) / float(records[:, 1].max())


def device_identity(device):
    """Physical identity of the GPU behind `device`: {'pci_bus_id', 'uuid', 'name', 'hostname'} (strings; None where the
    platform does not tell).  A rank's LOCAL device index says nothing about which GPU it drives (HIP_VISIBLE_DEVICES remaps):
    the multi-GPU bench line carries these per rank and refuses to print when two ranks sit on one physical device."""
    import socket
    from . import hostbind
    ident = {'pci_bus_id': None, 'uuid': None, 'name': None, 'hostname': socket.gethostname()}
    try:
        ident['pci_bus_id'] = hostbind.gpu_pci_bus_id(device)
    except Exception:
        pass
    try:
        props = torch.cuda.get_device_properties(device)
        ident['name'] = props.name
        u = getattr(props, 'uuid', None)
        ident['uuid'] = None if u is None else str(u)
    except Exception:
        pass
    return ident


def gather_objects(obj, dist=None):
    """One picklable object per rank -> list over ranks (identity without an initialised process group)."""
    if dist is None or not dist.is_initialized():
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


def shared_devices(identities):
    """identities: list of device_identity dicts (one per rank) -> list of rank groups that sit on ONE physical device (empty
    when every rank has its own).  Two ranks share a device when host and PCI bus id (or, without one, the UUID) coincide; a
    rank whose platform reports neither is its own group (nothing can be said)."""
    groups = {}
    for r, d in enumerate(identities):
        key = d.get('pci_bus_id') or d.get('uuid')
        if key is None:
            continue
        groups.setdefault((d.get('hostname'), key), []).append(r)
    return [g for g in groups.values() if len(g) > 1]
