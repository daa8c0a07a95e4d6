"""Drop-in replacements for the reference's ViSNet calculator seam
(/root/reference/src/Calculators/visnet_calculator.py):

    ViSNetModel            :22-75    in-process model, dl_potential_loader()
    ViSNetCalculator       :121-155  ASE-style whole-molecule calculator
    get_visnet_model       :184-204  factory used by DLBondedCalculator

Same names, argument meaning and return shapes; the network itself is the
hand-written HIP library behind include/vsn.h (no TorchScript, no autograd).
PyTorch is used only to hold the weights / IO tensors in HBM and for streams.
"""
from __future__ import annotations

import ctypes as C
import re

import numpy as np
import torch

from . import capi
from .fragment import FragmentData


def _as_numpy(v):
    if isinstance(v, torch.Tensor):
        return v.detach().cpu().numpy()
    return np.asarray(v)


class ViSNetEngine:
    """Thin owner of one vsn_handle (one device, one stream)."""

    def __init__(self, hparams: dict, state_dict: dict, device: str):
        if not isinstance(device, str) or not device.startswith("cuda"):
            raise RuntimeError(
                f"device={device!r}: the MI355X ViSNet calculator runs on 'cuda:<k>' (ROCm) devices only; "
                "there is no CPU path in this package"
            )
        self.device = torch.device(device)
        self.index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.hparams = dict(hparams)
        L = capi.lib()
        self._L = L
        for key, ok in (("rbf_type", tuple(capi.RBF)), ("activation", tuple(capi.ACTIVATION)),
                        ("attn_activation", tuple(capi.ACTIVATION)), ("model", ("ViSNetBlock",)),
                        ("output_model", ("Scalar",)), ("reduce_op", ("add", "mean"))):
            if hparams.get(key, ok[0]) not in ok:
                raise NotImplementedError(f"{key}={hparams.get(key)!r} is not built in the HIP path (supported: {ok})")
        prior = hparams.get("prior_model")
        if prior not in (None, "Atomref", False):
            raise NotImplementedError(f"prior_model={prior!r}")
        hp = capi.VsnHParams(
            hidden=int(hparams["embedding_dimension"]),
            num_layers=int(hparams["num_layers"]),
            num_rbf=int(hparams["num_rbf"]),
            num_heads=int(hparams["num_heads"]),
            lmax=int(hparams["lmax"]),
            max_z=int(hparams["max_z"]),
            max_num_neighbors=int(hparams["max_num_neighbors"]),
            vecnorm_type=capi.VECNORM[hparams["vecnorm_type"]],
            has_atomref=1 if prior == "Atomref" else 0,
            cutoff=float(hparams["cutoff"]),
            rbf_type=capi.RBF[hparams.get("rbf_type", "expnorm")],
            activation=capi.ACTIVATION[hparams.get("activation", "silu")],
            attn_activation=capi.ACTIVATION[hparams.get("attn_activation", "silu")],
        )
        self._h = C.c_void_p()
        rc = L.vsn_create(C.byref(self._h), C.byref(hp), self.index)
        self._check(rc)
        # weights live in HBM as PyTorch-ROCm tensors; the library packs its own fused copies
        self.weights = {}
        self.load_state_dict(state_dict)
        if hparams.get("reduce_op", "add") == "mean":  # visnet.py:146: per-fragment mean of the atomic terms
            self.set_option("reduce_mean", 1)

    def load_state_dict(self, state_dict: dict):
        """(re)load every tensor and re-pack (vsn_load_weight + vsn_finalize; a handle may be re-finalized with new
        values of the same shapes - every derived copy, the split-3 planes included, is rebuilt)"""
        L = self._L
        for name, val in state_dict.items():
            name = re.sub(r"^model\.", "", name)
            if name == "prior_model.initial_atomref":
                continue
            t = torch.as_tensor(_as_numpy(val), dtype=torch.float32).contiguous().to(self.device)
            self.weights[name] = t
            shape = (C.c_int64 * max(t.dim(), 1))(*(list(t.shape) or [1]))
            rc = L.vsn_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), shape, t.dim())
            self._check(rc)
        torch.cuda.synchronize(self.device)
        self._check(L.vsn_finalize(self._h))

    def _check(self, rc):
        if rc != 0:
            msg = self._L.vsn_last_error(self._h)
            raise RuntimeError(f"vsn error {rc}: {msg.decode() if msg else ''}")

    def set_option(self, key: str, value: int):
        self._check(self._L.vsn_set_option(self._h, key.encode(), int(value)))

    def forces_device(self, z: torch.Tensor, pos: torch.Tensor, start: np.ndarray, end: np.ndarray,
                      e_out: torch.Tensor, f_out: torch.Tensor, stream=None):
        """All tensors already in HBM; asynchronous on `stream` (torch stream or None=current)."""
        assert z.dtype == torch.int64 and pos.dtype == torch.float32 and z.is_cuda and pos.is_cuda
        assert pos.is_contiguous() and z.is_contiguous() and e_out.is_contiguous() and f_out.is_contiguous()
        start = np.ascontiguousarray(start, dtype=np.int64)
        end = np.ascontiguousarray(end, dtype=np.int64)
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        rc = self._L.vsn_forces(self._h, C.c_void_p(z.data_ptr()), C.c_void_p(pos.data_ptr()), capi.i64_ptr(start),
                                capi.i64_ptr(end), int(z.numel()), int(len(start)), C.c_void_p(e_out.data_ptr()),
                                C.c_void_p(f_out.data_ptr()), C.c_void_p(st.cuda_stream))
        self._check(rc)

    def profile_read(self):
        """-> {variant: dict(launches, ms, flops, bytes)} accumulated since set_option('profile', 1)."""
        out = (C.c_double * 16)()
        self._check(self._L.vsn_profile_read(self._h, out))
        names = ("k_gemm<128,128>", "k_gemm<64,64>", "k_gemm<128,32>", "k_gemm_group")
        return {names[v]: dict(launches=out[4 * v], ms=out[4 * v + 1], flops=out[4 * v + 2], bytes=out[4 * v + 3])
                for v in range(4)}

    def profile_read_scatter(self):
        """-> {kernel: dict(launches, ms, bytes)} of the bracketed scatter-path launches (forward edge attention and
        vector-message aggregation + node update), accumulated since set_option('profile', 1)."""
        out = (C.c_double * 8)()
        self._check(self._L.vsn_profile_read_scatter(self._h, out))
        names = ("k_edge_attn", "k_node_update")
        return {names[v]: dict(launches=out[4 * v], ms=out[4 * v + 1], bytes=out[4 * v + 2]) for v in range(2)}

    WALKS = ("k_edge_attn", "k_node_update", "k_bwd_hf1", "k_bwd_hf2", "k_bwd_attn_S", "k_bwd_norm_update")

    def profile_read_walks(self):
        """-> {kernel: dict(launches, ms, bytes)} of every timed node walk (forward scatter path + the reverse walks of
        single-protein sizes), accumulated since set_option('profile', 1); dispatch begin..end timestamps."""
        out = (C.c_double * (4 * len(self.WALKS)))()
        n = self._L.vsn_profile_read_walks(self._h, out, len(self.WALKS))
        if n < 0:
            self._check(n)
        return {self.WALKS[v]: dict(launches=out[4 * v], ms=out[4 * v + 1], bytes=out[4 * v + 2]) for v in range(n)}

    def profile_bracket_ms(self) -> float:
        """Average cost of an empty HIP-event bracket measured in the profiled calls (subtract it per launch)."""
        return float(self._L.vsn_profile_bracket_ms(self._h))

    def last_num_edges(self) -> int:
        return int(self._L.vsn_last_num_edges(self._h))

    @property
    def z_limit(self) -> int:
        """Atomic numbers must satisfy 0 <= z < z_limit (rows of the embedding tables and of the Atomref table)."""
        lim = int(self.hparams["max_z"])
        ar = self.weights.get("prior_model.atomref.weight")
        return min(lim, int(ar.numel())) if ar is not None and self.hparams.get("prior_model") == "Atomref" else lim

    def check_status(self):
        """Synchronises; raises IndexError if the last chunk saw an atomic number outside the tables (the kernels
        clamp the index and NaN-poison that chunk's outputs; nn.Embedding raises in the reference)."""
        st = int(self._L.vsn_last_status(self._h))
        if st < 0:
            self._check(st)
        if st & 1:
            raise IndexError(f"atomic number outside [0, {self.z_limit}) in the last fragment batch")

    def debug_read(self, name: str, layer: int = 0, dtype=np.float32, max_elems: int = 1 << 28) -> np.ndarray:
        # size query by reading into a generously sized buffer
        buf = np.empty(max_elems, dtype=dtype)
        n = self._L.vsn_debug_read(self._h, name.encode(), int(layer), C.c_void_p(buf.ctypes.data), int(max_elems))
        if n < 0:
            self._check(int(n))
        return buf[:n].copy()

    def gemm(self, A, Bt, C_out, bias=None, flags=0, stream=None):
        st = stream if stream is not None else torch.cuda.current_stream(self.device)
        M, K = A.shape
        Nc = Bt.shape[0]
        rc = self._L.vsn_gemm(self._h, C.c_void_p(A.data_ptr()), A.stride(0), C.c_void_p(Bt.data_ptr()),
                              Bt.stride(0), C.c_void_p(C_out.data_ptr()), C_out.stride(0),
                              C.c_void_p(bias.data_ptr() if bias is not None else 0), M, Nc, K, flags,
                              C.c_void_p(st.cuda_stream))
        self._check(rc)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._L.vsn_destroy(self._h)
                self._h = None
        except Exception:
            pass


class LoadedViSNet:
    """What `load_model` returns here: the checkpoint's hyper-parameters and tensors (the reference returns a
    scripted torch module; this package evaluates the network with its own kernels, so the 'model' is its weights)."""

    def __init__(self, hparams: dict, state_dict: dict):
        self.hparams, self.state = dict(hparams), dict(state_dict)

    def state_dict(self):
        return self.state

    def eval(self):
        return self

    def to(self, device):
        return self


def load_model(filepath, device="cpu"):
    """Mirror of ViSNet/model/visnet.py:73-93: reads the Lightning-style checkpoint - hyper-parameters from the
    file itself, 'model.' prefix stripped from the state_dict keys - and returns the loaded model object that
    `ViSNetModel(model, device)` takes."""
    ckpt = torch.load(filepath, map_location="cpu", weights_only=False)
    hp = ckpt["hyper_parameters"]
    sd = {re.sub(r"^model\.", "", k): v for k, v in ckpt["state_dict"].items()}
    return LoadedViSNet(hp, sd)


def load_checkpoint(filepath):
    m = load_model(filepath)
    return m.hparams, m.state


_ACT_NAMES = {"SiLU": "silu", "Swish": "swish", "ShiftedSoftplus": "ssp", "Tanh": "tanh", "Sigmoid": "sigmoid"}


def _act_name(layer_act):
    """activation name of a (possibly TorchScript-compiled) activation module"""
    cls = getattr(layer_act, "original_name", None) or type(layer_act).__name__
    if cls not in _ACT_NAMES:
        raise TypeError(f"hparams_of_module: unknown activation module {cls}")
    return _ACT_NAMES[cls]


def hparams_of_module(model) -> dict:
    """Hyper-parameters of a torch ViSNet module built by the reference's create_model (visnet.py:14-70), read off
    the attributes ViSNetBlock keeps (visnet_block.py:40-55) - so that a reference module can be handed to
    `ViSNetModel(model, device)` exactly where the reference constructs its own.  The reference's load_model returns
    `torch.jit.script(model)` (visnet.py:92): a scripted module keeps the sub-module tree, the state_dict and
    `original_name`, but not every plain Python attribute, so sizes fall back to the state_dict shapes and the
    remaining scalars must be readable (TorchScript keeps int/float/str attributes that the scripted code uses)."""
    rep = model.representation_model
    sd = model.state_dict()
    layers = rep.vis_mp_layers
    layer = layers[0]

    def attr(name, default=None):
        v = getattr(rep, name, default)
        if v is None:
            raise TypeError(f"hparams_of_module: the module does not expose `{name}` (a scripted module that dropped "
                            "it): pass the checkpoint through load_model() instead")
        return v

    H = int(sd["representation_model.embedding.weight"].shape[1])
    L = len(layers)
    R = int(sd["representation_model.neighbor_embedding.distance_proj.weight"].shape[1])
    max_z = int(sd["representation_model.embedding.weight"].shape[0])
    act = getattr(rep, "activation", None)
    attn = getattr(rep, "attn_activation", None)
    prior = getattr(model, "prior_model", None)
    return dict(
        model="ViSNetBlock", output_model="Scalar", reduce_op=getattr(model, "reduce_op", "add"),
        embedding_dimension=H, num_layers=L, num_rbf=R,
        num_heads=int(attr("num_heads")), lmax=int(attr("lmax")), max_z=max_z, cutoff=float(attr("cutoff")),
        max_num_neighbors=int(attr("max_num_neighbors")), vecnorm_type=attr("vecnorm_type"), rbf_type=attr("rbf_type"),
        activation=act if isinstance(act, str) else _act_name(layer.act),
        attn_activation=attn if isinstance(attn, str) else _act_name(layer.attn_activation),
        prior_model="Atomref" if prior is not None else None,
    )


class ViSNetModel:
    r"""Energy and forces of a fragment batch from the ViSNet potential
    (mirror of Calculators/visnet_calculator.py:22-75)."""

    implemented_properties = ["energy", "forces"]

    def __init__(self, model, state_dict=None, device="cuda:0"):
        """`ViSNetModel(model, device=...)` like the reference (:35): `model` is what `load_model` returned, or a torch
        ViSNet module built by the reference's create_model.  The three-argument form `(hparams, state_dict, device)`
        builds one from in-memory weights."""
        if isinstance(state_dict, (str, torch.device)):
            # the reference's positional form ViSNetModel(model, "cuda:0") (visnet_calculator.py:35)
            device, state_dict = state_dict, None
        if state_dict is not None:
            hparams, sd = model, state_dict
        elif isinstance(model, LoadedViSNet):
            hparams, sd = model.hparams, model.state
        elif isinstance(model, torch.nn.Module):
            hparams, sd = hparams_of_module(model), model.state_dict()
        else:
            raise TypeError(f"ViSNetModel: cannot take a model of type {type(model).__name__}")
        self.model = model
        self.device = device
        self.engine = ViSNetEngine(hparams, sd, device)
        self.stream = torch.cuda.Stream(device=device)

    def collate(self, frag: FragmentData):
        z = torch.as_tensor(np.ascontiguousarray(frag.z, dtype=np.int64)).to(self.device, non_blocking=True)
        pos = torch.as_tensor(np.ascontiguousarray(frag.pos, dtype=np.float32)).to(self.device, non_blocking=True)
        return dict(z=z, pos=pos, start=np.asarray(frag.start, dtype=np.int64),
                    end=np.asarray(frag.end, dtype=np.int64))

    def _io_buffers(self, N: int, B: int):
        """Persistent staging for the host seam: pinned host buffers and device buffers that grow on demand, the two
        outputs in ONE device buffer [e (cap_b) | f (3 cap_n)] so that a call costs two H2D copies, one D2H copy and one
        stream synchronisation (the reference's seam pays three H2D and two blocking D2H per call,
        visnet_calculator.py:47-63)."""
        io = getattr(self, "_io", None)
        if io is None or io["cap_n"] < N or io["cap_b"] < B:
            cap_n = max(N, 2 * (io["cap_n"] if io else 0), 64)
            cap_b = max(B, 2 * (io["cap_b"] if io else 0), 8)
            dev = self.device
            io = dict(
                cap_n=cap_n, cap_b=cap_b,
                pin_z=torch.empty(cap_n, dtype=torch.int64).pin_memory(),
                pin_pos=torch.empty(cap_n, 3, dtype=torch.float32).pin_memory(),
                dev_z=torch.empty(cap_n, dtype=torch.int64, device=dev),
                dev_pos=torch.empty(cap_n, 3, dtype=torch.float32, device=dev),
                dev_out=torch.empty(cap_b + 3 * cap_n, dtype=torch.float32, device=dev),
                pin_out=torch.empty(cap_b + 3 * cap_n, dtype=torch.float32).pin_memory(),
            )
            self._io = io
        return io

    def dl_potential_loader(self, frag_data: FragmentData):
        """-> (e float32 [B_nonempty, 1], f float32 [N, 3]) as numpy, like the reference (:54-63)."""
        with self._call_lock():
            return self._dl_potential_loader(frag_data)

    def _call_lock(self):
        """One handle = one device, one stream, one workspace and one set of staging buffers: concurrent callers of the
        SAME handle serialise here.  The reference's factory hands the same model to every entry of
        DeviceStrategy.get_bonded_devices() that names the same device (`_local_calc`, visnet_calculator.py:184-204) and
        its DLBondedCalculator drives each entry from its own thread (bonded.py:75-77)."""
        lock = self.__dict__.get("_lock")
        if lock is None:
            import threading

            lock = self.__dict__.setdefault("_lock", threading.Lock())
        return lock

    def _dl_potential_loader(self, frag_data: FragmentData):
        z_host = np.ascontiguousarray(frag_data.z, dtype=np.int64)
        if z_host.size and (z_host.min() < 0 or z_host.max() >= self.engine.z_limit):
            # nn.Embedding / Atomref raise for an index outside their table (visnet_block.py:110, priors.py:86-87)
            raise IndexError(f"atomic number outside [0, {self.engine.z_limit}) in FragmentData.z")
        pos_host = np.ascontiguousarray(frag_data.pos, dtype=np.float32).reshape(-1, 3)
        start = np.asarray(frag_data.start, dtype=np.int64)
        end = np.asarray(frag_data.end, dtype=np.int64)
        N, B = int(z_host.size), int(len(start))
        io = self._io_buffers(N, B)
        cb = io["cap_b"]
        io["pin_z"].numpy()[:N] = z_host
        io["pin_pos"].numpy()[:N] = pos_host
        with torch.cuda.stream(self.stream):
            z = io["dev_z"][:N]
            pos = io["dev_pos"][:N]
            z.copy_(io["pin_z"][:N], non_blocking=True)
            pos.copy_(io["pin_pos"][:N], non_blocking=True)
            e = io["dev_out"][:B]
            f = io["dev_out"][cb: cb + 3 * N].view(N, 3)
            self.engine.forces_device(z, pos, start, end, e, f, stream=self.stream)
            io["pin_out"][: cb + 3 * N].copy_(io["dev_out"][: cb + 3 * N], non_blocking=True)
        self.stream.synchronize()
        out = io["pin_out"].numpy()
        nonempty = (end - start) > 0
        e_np = out[:B][nonempty].reshape(-1, 1).copy()
        f_np = out[cb: cb + 3 * N].reshape(-1, 3).copy()
        return e_np, f_np

    @classmethod
    def from_file(cls, **kwargs):
        if "model_path" not in kwargs:
            raise ValueError("model_path must be provided")
        model = load_model(kwargs["model_path"])
        return cls(model, device=kwargs.get("device", "cuda:0"))


class ViSNetCalculator:
    r"""Whole-molecule mode (`--mode visnet`): one fragment = the whole system (mirror of
    Calculators/visnet_calculator.py:121-155, same constructor).  ASE's Calculator base class is not importable in
    this package's test environment, so the class implements the `calculate(atoms, properties, system_changes)` /
    `.results` / `implemented_properties` protocol stand-alone; with ASE present, list `ase.calculators.calculator.
    Calculator` as a second base - nothing else changes."""

    implemented_properties = ["energy", "forces"]

    def __init__(self, ckpt_path: str = None, ckpt_type: str = None, is_root_calc=True, model: ViSNetModel = None,
                 **kwargs):
        import os.path as osp

        from .device_strategy import DeviceStrategy

        self.ckpt_path, self.ckpt_type, self.is_root_calc = ckpt_path, ckpt_type, is_root_calc
        self.results = {}
        if model is not None:  # embedding aid: an already constructed seam object
            self.model, self.device = model, model.device
            return
        model_path = osp.join(self.ckpt_path, f"visnet-uni-{self.ckpt_type}.ckpt")
        self.device = DeviceStrategy.get_bonded_devices()[0]
        self.model = get_visnet_model(model_path, self.device)

    def calculate(self, atoms, properties=("energy", "forces"), system_changes=None):
        n = len(atoms)
        data = FragmentData(
            np.asarray(atoms.numbers),
            np.asarray(atoms.positions, dtype=np.float32),
            np.array([0], dtype=np.int64),
            np.array([n], dtype=np.int64),
            np.zeros((n,), dtype=np.int64),
        )
        e, f = self.model.dl_potential_loader(data)
        self.results = {"energy": e, "forces": f}


_local_calc: dict = {}


def get_visnet_model(model_path: str, device: str):
    """Factory with the reference's signature (:184-204).  The reference keeps one
    GPU model in the master process and spawns pickle-over-socket workers for the
    other GPUs; here every 'cuda:k' gets its own in-process handle (handles on
    different devices run concurrently from different threads)."""
    if device == "cpu":
        raise RuntimeError("get_visnet_model(device='cpu'): this package is the MI355X path; no CPU model exists")
    sig = f"{device}-{model_path}"
    if sig not in _local_calc:
        _local_calc[sig] = ViSNetModel.from_file(model_path=model_path, device=device)
    return _local_calc[sig]
