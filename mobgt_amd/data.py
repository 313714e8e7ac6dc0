"""Device-side batch construction: raw trajectories -> padded `Batch1` on the GPU in one pass.

This is the default data path of the trainer / bench.  It replaces the per-sample
`wrapper.preprocess_item` + `algos` calls and the per-batch `collator_*` of the reference
(`wrapper.py:25-102`, `collator.py:310-458`) with: one host-side packing of the raw dicts
(`gen_pickles.py:820-832` format) into padded arrays, one H2D copy each, `mobgt_spd_batched` for
SPD / path / first-D-hop edge features / degrees, and a gather from a precomputed distance-bin table for
`poi_pos`.  Index tensors stay narrow on the device (int16 / uint8 instead of the reference's int64):
the kernels are templated on the index type, and `.long()` views give the reference dtypes when needed.
Every field equals what the reference pipeline produces for the same trajectories (tests/test_gpu_data.py).
"""
import numpy as np
import torch

from . import ops
from .collator import Batch1, freedman_diaconis_bins


def make_bin_table(distance, dtype=np.int16):
    """(P+1)x(P+1) distance matrix -> (num_bins, bin ids): bin[a,b] = np.digitize(distance[a,b], edges) with the
    Freedman-Diaconis edges of collator.py:430-433.  Computed once instead of per batch."""
    d = np.asarray(distance)
    dm = np.delete(np.delete(d, 0, axis=0), 0, axis=1)
    num_bins, edges = freedman_diaconis_bins(dm - dm.min(), True)
    return num_bins, edges, np.digitize(d, edges).astype(dtype)


class DeviceBatch1(Batch1):
    """`Batch1` whose model-unused fields (adj, adj1, attn_edge_type, feature_matrix) are derived on first
    access from the packed counts, so the hot path does not pay for them."""
    _fields = ("idx", "attn_bias", "rel_pos", "in_degree", "out_degree", "x", "edge_input", "y", "time", "time_normal",
               "user", "cat", "poi_pos")

    def __init__(self, counts, n_nodes, **kw):
        for f in self._fields:
            setattr(self, f, kw[f])
        self._feature_matrix = None
        self._counts, self._n_nodes = counts, n_nodes

    def to(self, device):
        for f in self._fields:
            setattr(self, f, getattr(self, f).to(device))
        self._counts, self._n_nodes = self._counts.to(device), self._n_nodes.to(device)
        return self

    @property
    def adj1(self):                                                   # wrapper.py:73-76, collator.py:386
        return self._counts != 0

    @property
    def adj(self):                                                    # wrapper.py:67-81, collator.py:384
        G, N = self._counts.shape[:2]
        a = torch.zeros(G, N + 1, N + 1, dtype=torch.bool, device=self._counts.device)
        a[:, :N, :N] = self._counts != 0
        ar = torch.arange(N + 1, device=a.device)
        n = self._n_nodes.long().view(G, 1)
        inside = ar.view(1, -1) <= n                                  # positions 0..n of each graph
        tok = ar.view(1, -1) == n                                     # the virtual token sits at index n
        a |= tok.unsqueeze(2) & inside.unsqueeze(1)
        a |= inside.unsqueeze(2) & tok.unsqueeze(1)
        return a

    @property
    def attn_edge_type(self):                                         # wrapper.py:49-53, collator.py:412-417
        G, N = self._counts.shape[:2]
        t = torch.zeros(G, N + 1, N + 1, 1, dtype=torch.long, device=self._counts.device)
        c = self._counts.long()
        t[:, :N, :N, 0] = torch.where(c != 0, c + 2, c)
        return t


class DeviceCollator:
    """callable(list of raw trajectory dicts) -> DeviceBatch1 on `device`.
    Mirrors `partial(collator_foursquare, max_node=30000, multi_hop_max_dist=D, rel_pos_max=R)` (data.py:289-294)
    applied to `preprocess_item`-ed items."""

    def __init__(self, device, bin_table=None, multi_hop_max_dist=20, rel_pos_max=1024, max_node=30000, coords=None,
                 bin_edges=None):
        """`bin_table`: (P+1) x (P+1) precomputed distance-bin ids (make_bin_table).  For universes where that table
        cannot exist (P = 100 000), `coords` [(P+1), 2] lat / lon + `bin_edges`: poi_pos = digitize(haversine) is then
        evaluated for the batch's pairs on the device."""
        self.device = torch.device(device)
        self.D = int(multi_hop_max_dist)
        self.rel_pos_max = int(rel_pos_max)
        self.max_node = int(max_node)
        self.bin_table = None
        if bin_table is not None:
            self.bin_table = torch.as_tensor(bin_table).to(self.device)
        self.coords = self.bin_edges = None
        if coords is not None:
            self.coords = torch.as_tensor(np.radians(np.asarray(coords, dtype=np.float64))).to(self.device)
            self.bin_edges = torch.as_tensor(np.asarray(bin_edges, dtype=np.float64)).to(self.device)

    def pack_host(self, trajs, idx0=0, n_pad=None, out=None):
        """Raw dicts -> padded numpy arrays (pinned-memory friendly); no graph algorithm runs on the host.
        `n_pad`: pad to this many nodes instead of the batch maximum (bucketed shapes: the extra positions are padding
        exactly like the collator's own, collator.py:11-101); `out`: pre-allocated arrays to fill (RawLayout.views)."""
        trajs = [t for t in trajs if t is not None and len(t["node_name"]) <= self.max_node]
        G = len(trajs)
        n = np.array([len(t["node_name"]) for t in trajs], dtype=np.int32)
        N = int(n.max()) if n_pad is None else int(n_pad)
        if N < int(n.max()):
            raise ValueError(f"n_pad {N} is smaller than the longest trajectory ({int(n.max())})")
        if out is not None:
            for k in ("counts", "x", "time", "cat", "time_normal"):
                out[k][...] = 0
            counts, x, time, cat, time_normal = out["counts"], out["x"], out["time"], out["cat"], out["time_normal"]
            assert counts.shape == (G, N, N), (counts.shape, G, N)
        else:
            counts = np.zeros((G, N, N), dtype=np.int32)
            x = np.zeros((G, N, 1), dtype=np.int32)
            time = np.zeros((G, N, 1), dtype=np.int32)
            cat = np.zeros((G, N, 1), dtype=np.int32)
            time_normal = np.zeros((G, N, 1), dtype=np.float32)
        for g, t in enumerate(trajs):
            k = n[g]
            counts[g, :k, :k] = t["edge_type"]
            x[g, :k, 0] = t["node_name"]                              # wrapper +1 then collator -1 (App. A)
            time[g, :k, 0] = t["time"]
            cat[g, :k, 0] = t["cat"]
            time_normal[g, :k, 0] = t["time_normal"]
        user = np.array([[int(t["user"][0]) + 1] for t in trajs], dtype=np.int32)       # wrapper.py:39
        y = np.array([int(t["target"][0]) for t in trajs], dtype=np.int64)              # collator.py:367
        idx = np.arange(idx0, idx0 + G, dtype=np.int64) if np.isscalar(idx0) else np.asarray(idx0, dtype=np.int64)
        if out is not None:
            out["n_nodes"][...], out["user"][...], out["y"][...], out["idx"][...] = n, user, y, idx
            return out
        return dict(counts=counts, n_nodes=n, x=x, time=time, cat=cat, time_normal=time_normal, user=user, y=y, idx=idx)

    def __call__(self, trajs, idx0=0, n_pad=None):
        h = self.pack_host(trajs, idx0, n_pad=n_pad)
        d = {k: torch.from_numpy(v).to(self.device, non_blocking=True) for k, v in h.items()}
        return self.finish(d)

    def can_finish_into(self):
        return self.coords is None and (self.bin_table is None or (self.bin_table.dtype == torch.int16 and self.bin_table.is_contiguous()))

    def finish_into(self, v, work=None):
        """`finish` writing into the pre-allocated views of a BatchLayout buffer (raw fields already in place): two / three
        launches on the CURRENT stream, no allocation -- train.EpochLoop runs it on its copy stream for the next batch."""
        from . import _lib
        from .ops import _p, _stream
        G, N = v["counts"].shape[:2]
        lib = _lib.lib()
        if work is None:
            work = torch.empty(int(lib.mobgt_spd_workspace_bytes(G, N)), dtype=torch.uint8, device=self.device)
        _lib.check(lib.mobgt_spd_batched(_p(v["counts"]), _p(v["n_nodes"]), _p(v["spd"]), _p(v["path"]), _p(v["rel_pos"]),
                                         _p(v["edge_input"]), _p(v["in_degree"]), _p(v["out_degree"]), _p(work), G, N, self.D,
                                         _stream()), "mobgt_spd_batched")
        bt = self.bin_table
        _lib.check(lib.mobgt_collate_finish(_p(v["x"]), _p(v["n_nodes"]), _p(v["spd"]), _p(bt), bt.shape[1] if bt is not None else 0,
                                            self.rel_pos_max, _p(v["attn_bias"]), _p(v["poi_pos"]), G, N, _stream()),
                   "mobgt_collate_finish")
        return work

    @staticmethod
    def batch_from_views(v):
        return DeviceBatch1(v["counts"], v["n_nodes"], idx=v["idx"], attn_bias=v["attn_bias"], rel_pos=v["rel_pos"],
                            in_degree=v["in_degree"], out_degree=v["out_degree"], x=v["x"], edge_input=v["edge_input"],
                            y=v["y"], time=v["time"], time_normal=v["time_normal"], user=v["user"], cat=v["cat"],
                            poi_pos=v["poi_pos"])

    def finish(self, d):
        counts, n_nodes = d["counts"], d["n_nodes"]
        G, N = counts.shape[:2]
        T = N + 1
        sp = ops.spd_batched(counts, n_nodes, self.D)
        x = d["x"]
        if (self.coords is None and x.dtype == torch.int32 and n_nodes.dtype == torch.int32 and x.is_contiguous()
                and (self.bin_table is None or (self.bin_table.dtype == torch.int16 and self.bin_table.is_contiguous()))):
            # padding mask, rel_pos_max cut and the distance-bin gather in one launch (mobgt_collate_finish)
            from . import _lib
            from .ops import _p, _stream
            attn_bias = torch.empty(G, T, T, device=self.device)
            poi_pos = torch.empty(G, N, N, dtype=torch.int16, device=self.device)
            bt = self.bin_table
            _lib.check(_lib.lib().mobgt_collate_finish(_p(x), _p(n_nodes), _p(sp["spd"]), _p(bt), bt.shape[1] if bt is not None else 0,
                                                       self.rel_pos_max, _p(attn_bias), _p(poi_pos), G, N, _stream()),
                       "mobgt_collate_finish")
            return DeviceBatch1(counts, n_nodes, idx=d["idx"], attn_bias=attn_bias, rel_pos=sp["rel_pos"],
                                in_degree=sp["in_degree"], out_degree=sp["out_degree"], x=x, edge_input=sp["edge_input"],
                                y=d["y"], time=d["time"], time_normal=d["time_normal"], user=d["user"], cat=d["cat"],
                                poi_pos=poi_pos)
        ar = torch.arange(T, device=self.device)
        real_tok = ar.view(1, T) <= n_nodes.view(G, 1)                # token 0 + n real nodes
        attn_bias = torch.zeros(G, T, T, device=self.device)
        attn_bias.masked_fill_(~real_tok.view(G, 1, T), float("-inf"))              # collator.py:57-64
        if self.rel_pos_max <= 510:                                                  # collator.py:354-358
            far = sp["spd"] >= self.rel_pos_max
            attn_bias[:, 1:, 1:].masked_fill_(far, float("-inf"))
        if self.bin_table is not None:
            xi = x[:, :, 0].long()
            poi_pos = self.bin_table[xi.unsqueeze(2), xi.unsqueeze(1)]
            real = xi != 0
            poi_pos = torch.where(real.unsqueeze(2) & real.unsqueeze(1), poi_pos, torch.zeros_like(poi_pos))
        elif self.coords is not None:
            xi = x[:, :, 0].long()
            ll = self.coords[xi]                                                     # [G,N,2] radians, float64
            lat1, lon1, lat2, lon2 = ll[:, :, None, 0], ll[:, :, None, 1], ll[:, None, :, 0], ll[:, None, :, 1]
            h = torch.sin((lat2 - lat1) / 2) ** 2 + torch.cos(lat1) * torch.cos(lat2) * torch.sin((lon2 - lon1) / 2) ** 2
            dist = 2 * 6371.0 * torch.asin(torch.sqrt(h.clamp(0.0, 1.0)))          # synth.haversine_km
            poi_pos = torch.bucketize(dist, self.bin_edges, right=True).to(torch.int16)   # == np.digitize(dist, edges)
            real = xi != 0
            poi_pos = torch.where(real.unsqueeze(2) & real.unsqueeze(1), poi_pos, torch.zeros_like(poi_pos))
        else:
            poi_pos = torch.zeros(G, N, N, dtype=torch.int16, device=self.device)
        return DeviceBatch1(counts, n_nodes, idx=d["idx"], attn_bias=attn_bias, rel_pos=sp["rel_pos"],
                            in_degree=sp["in_degree"], out_degree=sp["out_degree"], x=x, edge_input=sp["edge_input"],
                            y=d["y"], time=d["time"], time_normal=d["time_normal"], user=d["user"], cat=d["cat"],
                            poi_pos=poi_pos)


# ---- bucketed shapes for a loop over FRESH batches (train.TrainStep.run_epoch) ------------------------------------------------
# A hipGraph has static shapes, and the reference feeds a new batch every step (data.py:282-295): the padded node count of
# a batch is therefore rounded UP to one of a few sizes and one step graph is kept per (G, size).  Rounding up only adds
# padding positions, which the collator already produces for every graph shorter than the batch maximum (-inf key columns,
# zero indices): logits, loss and gradients do not change (tests/test_gpu_loop.py).
# (round 4: steps of 1/16 .. 1/8 of the size instead of round 3's 1/3 .. 1/2 (8, 16, 24, 32, 48, 64, 96, ...): at most ~6-12 % of
#  padding in N instead of ~33 %.  A bucket costs one captured step graph, taken the first time it occurs.  Measured on the
#  S-FSQ pool: fresh-batch loop 0.784 -> 0.700 ms per step with steps of 1/8 .. 1/4)
BUCKETS = (tuple(range(4, 65, 4)) + tuple(range(72, 129, 8)) + tuple(range(144, 257, 16)) + tuple(range(288, 513, 32))
           + tuple(range(576, 1025, 64)))


def balanced_batches(lengths, world_size, batch_size, epoch=0, seed=0, shuffle=True, buckets=BUCKETS, window=None):
    """Length-balanced dealing of one epoch over `world_size` ranks ("load imbalance across ranks": node
    counts run from 1 to 814 on Gowalla, a step costs what its padded N costs, and a synchronous step lasts as long as its
    slowest rank): -> steps[j][r] = sample ids of rank r in step j, the same list on every rank.

    * the epoch's samples are exactly DistributedSampler's (`shard_indices`): the permutation seeded by seed + epoch, padded by
      wrap-around to a multiple of world_size -- the union over ranks is the same multiset, every rank takes the same number of
      samples and of steps, the last step is the short one on all ranks alike (drop_last=False, data.py:282-295);
    * inside it the samples are ordered by the BUCKET their node count pads to (stable: samples of one bucket keep their random
      order, so a batch's composition stays random) and cut into steps of world_size x batch_size; rank r takes the r-th run of a
      step -- neighbouring runs of one sorted sequence, i.e. the same or the neighbouring bucket on all ranks;
    * the steps are then visited in an order drawn from seed + epoch + 1 (sizes mixed over time: the learning-rate schedule
      never sees 'all short graphs first');
    * `window` (steps; None = the whole epoch): the sort runs inside consecutive windows of window x world_size x batch_size
      samples of the permuted epoch only -- every window is a uniform random sample of the dataset, so a step's batches are
      length-homogeneous only with respect to the ~window x world x batch samples around them, and the long graphs of an epoch are
      spread over its windows instead of sharing one step (whole-epoch sorting changes SGD's batch statistics against
      the reference's i.i.d. DistributedSampler batches; `train.EpochLoop` therefore uses a bounded window by default)."""
    n = len(lengths)
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n, generator=g).tolist()
    else:
        order = list(range(n))
    total = (n + world_size - 1) // world_size * world_size
    order += order[: total - len(order)]
    per_step = world_size * batch_size
    span = total if window is None else max(1, int(window)) * per_step
    steps = []
    for w0 in range(0, total, span):
        keyed = sorted(order[w0:w0 + span], key=lambda i: bucket_nodes(int(lengths[i]), buckets))          # (stable)
        for s in range(0, len(keyed), per_step):
            chunk = keyed[s:s + per_step]
            m = len(chunk) // world_size                                                 # (total is a multiple of world_size)
            steps.append([chunk[r * m:(r + 1) * m] for r in range(world_size)])
    # (windows: only the epoch's LAST step can be short -- every window but the last holds whole steps)
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch + 1)
        steps = [steps[i] for i in torch.randperm(len(steps), generator=g).tolist()]
    return steps


def bucket_nodes(n, buckets=BUCKETS):
    """Smallest bucket >= n (beyond the largest: the next multiple of 256)."""
    for b in buckets:
        if n <= b:
            return b
    return (n + 255) // 256 * 256


class RawLayout:
    """The raw (un-collated) arrays of one (G, N) bucket packed into ONE byte buffer, so that a step's input is one
    host-to-device copy: 8-byte fields first, every field 16-byte aligned."""
    FIELDS = (("y", np.int64, lambda G, N: (G,)), ("idx", np.int64, lambda G, N: (G,)),
              ("counts", np.int32, lambda G, N: (G, N, N)), ("x", np.int32, lambda G, N: (G, N, 1)),
              ("time", np.int32, lambda G, N: (G, N, 1)), ("cat", np.int32, lambda G, N: (G, N, 1)),
              ("time_normal", np.float32, lambda G, N: (G, N, 1)), ("n_nodes", np.int32, lambda G, N: (G,)),
              ("user", np.int32, lambda G, N: (G, 1)))

    def __init__(self, G, N):
        self.G, self.N = int(G), int(N)
        self.offsets, off = {}, 0
        for name, dt, shape in self.FIELDS:
            shp = shape(self.G, self.N)
            nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
            self.offsets[name] = (off, nbytes, dt, shp)
            off = (off + nbytes + 15) // 16 * 16
        self.nbytes = off

    def views_np(self, buf):
        """numpy views of a host uint8 array (e.g. `pinned_tensor.numpy()`)"""
        return {k: buf[o:o + n].view(dt).reshape(shp) for k, (o, n, dt, shp) in self.offsets.items()}

    def views_torch(self, buf):
        """typed torch views of a (device) uint8 tensor"""
        tdt = {np.int64: torch.int64, np.int32: torch.int32, np.float32: torch.float32}
        return {k: buf[o:o + n].view(tdt[dt]).view(*shp) for k, (o, n, dt, shp) in self.offsets.items()}


class BatchLayout(RawLayout):
    """RawLayout + everything the device collate derives from it, in one byte buffer: [raw fields | derived fields |
    scratch].  A staging copy of this buffer is filled on a side stream (H2D of the raw part, then the collate kernels)
    while the previous step runs; ONE device-to-device copy of [raw | derived] then refreshes the static buffer the bucket's
    step graph reads."""

    def __init__(self, G, N, D):
        super().__init__(G, N)
        self.D = int(D)
        self.raw_bytes = self.nbytes
        T = self.N + 1
        off = self.nbytes
        derived = (("attn_bias", np.float32, (self.G, T, T)), ("rel_pos", np.int16, (self.G, self.N, self.N)),
                   ("poi_pos", np.int16, (self.G, self.N, self.N)), ("edge_input", np.uint8, (self.G, self.N, self.N, self.D, 1)),
                   ("in_degree", np.int16, (self.G, self.N)), ("out_degree", np.int16, (self.G, self.N)))
        scratch = (("spd", np.int16, (self.G, self.N, self.N)), ("path", np.int16, (self.G, self.N, self.N)))
        for group in (derived, scratch):
            for name, dt, shp in group:
                nbytes = int(np.prod(shp)) * np.dtype(dt).itemsize
                self.offsets[name] = (off, nbytes, dt, shp)
                off = (off + nbytes + 15) // 16 * 16
            if group is derived:
                self.copy_bytes = off                      # [raw | derived]: what a step's graph reads
        self.nbytes = off

    def views_torch(self, buf):
        tdt = {np.int64: torch.int64, np.int32: torch.int32, np.float32: torch.float32, np.int16: torch.int16, np.uint8: torch.uint8}
        return {k: buf[o:o + n].view(tdt[dt]).view(*shp) for k, (o, n, dt, shp) in self.offsets.items()}

    def views_np(self, buf):
        """numpy views of the RAW part of a host uint8 array (a pinned staging buffer of raw_bytes)"""
        return {k: buf[o:o + n].view(dt).reshape(shp) for k, (o, n, dt, shp) in self.offsets.items() if o + n <= self.raw_bytes}


def shard_indices(n_samples, rank, world_size, epoch=0, seed=0, shuffle=True):
    """torch DistributedSampler semantics (what Lightning's DDP uses, entry.py:141): permutation seeded by
    seed+epoch, padded by wrap-around to a multiple of world_size, strided by rank."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        order = torch.randperm(n_samples, generator=g).tolist()
    else:
        order = list(range(n_samples))
    total = (n_samples + world_size - 1) // world_size * world_size
    order += order[: total - len(order)]
    return order[rank:total:world_size]
