"""CPU affinity of a rank: the cores of its GPU's NUMA node.

Eight ranks of one node each run a main thread and a producer thread (``HipModel.predict_dataset``, ``trainer.fit``); left to
float over all cores they share caches with seven strangers and cross the socket to reach their GPU's PCIe root.  A rank
therefore pins itself -- before its first GPU call -- to the cores of the NUMA node its device hangs off
(``/sys/class/drm/card*/device/numa_node``), and the ranks whose devices share a node split that node's cores between them.
Nothing here touches the GPU; every step is best-effort (an unknown topology leaves the affinity alone) and
``SCANN_NO_AFFINITY=1`` turns it off.
"""
from __future__ import annotations

import glob
import os


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def parse_cpulist(text):
    """'0-3,8,10-11' -> {0, 1, 2, 3, 8, 10, 11}"""
    cpus = set()
    for part in (text or "").split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            lo, hi = part.split("-", 1)
            cpus.update(range(int(lo), int(hi) + 1))
        else:
            cpus.add(int(part))
    return cpus


def amd_gpus(sysfs="/sys"):
    """[(pci address, numa node)] of the AMD display devices in PCI order -- the order HIP enumerates them in (a numa node of -1,
    as single-socket hosts report, is returned as it is)."""
    seen = {}
    for dev in glob.glob(os.path.join(sysfs, "class", "drm", "card*", "device")):
        if "-" in os.path.basename(os.path.dirname(dev)):  # card0-DP-1 and friends: connectors, not devices
            continue
        if (_read(os.path.join(dev, "vendor")) or "").lower() != "0x1002":
            continue
        node = _read(os.path.join(dev, "numa_node"))
        if node is None:
            continue
        addr = os.path.basename(os.path.realpath(dev))
        seen[addr] = int(node)
    return sorted(seen.items())


def _visible(n_total):
    """HIP's view of the devices: HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES as a list of indices (anything else: all of them)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v:
            try:
                idx = [int(x) for x in v.split(",") if x.strip() != ""]
            except ValueError:
                return list(range(n_total))
            return [i for i in idx if 0 <= i < n_total]
    return list(range(n_total))


def cpus_for_device(device, sysfs="/sys", allowed=None):
    """The cores rank-on-``device`` should run on: its NUMA node's cores (inside ``allowed``), split evenly with the other visible
    devices of the same node.  None when the topology does not say (no such device, node -1, empty intersection)."""
    gpus = amd_gpus(sysfs)
    vis = _visible(len(gpus))
    if not 0 <= device < len(vis):
        return None
    node = gpus[vis[device]][1]
    if node < 0:
        return None
    cpus = parse_cpulist(_read(os.path.join(sysfs, "devices", "system", "node", "node%d" % node, "cpulist")))
    if allowed is not None:
        cpus &= set(allowed)
    if not cpus:
        return None
    mates = [d for d in range(len(vis)) if gpus[vis[d]][1] == node]  # devices sharing the node, in device order
    cpus = sorted(cpus)
    share = len(cpus) // len(mates)
    if share < 2:  # too few cores to split: the whole node for every one of them
        return set(cpus)
    k = mates.index(device)
    return set(cpus[k * share:(k + 1) * share] if k < len(mates) - 1 else cpus[k * share:])


def pin_to_device(device, sysfs="/sys"):
    """Pin the CALLING THREAD (threads it starts later inherit the mask; threads that already exist keep theirs) to
    ``cpus_for_device``; returns the set or None.  Call before any worker thread is started.  With both HIP_VISIBLE_DEVICES and
    ROCR_VISIBLE_DEVICES set the device index cannot be mapped back to a physical card from here (HIP's list indexes ROCR's): no pinning."""
    if os.environ.get("SCANN_NO_AFFINITY") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    if os.environ.get("HIP_VISIBLE_DEVICES") and os.environ.get("ROCR_VISIBLE_DEVICES"):
        return None
    try:
        cpus = cpus_for_device(int(device), sysfs, allowed=os.sched_getaffinity(0))
        if cpus:
            os.sched_setaffinity(0, cpus)
        return cpus
    except (OSError, ValueError):
        return None
