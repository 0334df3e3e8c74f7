"""ctypes binding of libugsm.so (include/ugsm.h) -- the only native entry into the product.

There is no CPU fallback: if the HIP library is missing or no device is present every
compute call raises (UgsmError).  The pure-host geometry calls work without a GPU.

libugsm_dev.so (include/ugsm_dev.h) is the product's sources plus csrc/dev/: kernel_path 1 (one kernel per reference stage), round 1's
LDS-tiled K-cost (march_min_pixels < 0) and the probe entry points.  Tests and tools reach it through load(dev=True) /
Context(dev=True); a Context that asks for kernel_path 1 or march_min_pixels < 0 gets it by itself.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libugsm.so")
DEV_LIB_PATH = os.path.join(_HERE, "libugsm_dev.so")
UGSM_MAX_BATCH = 16

UGSM_OK = 0
UGSM_ERR_BAD_ARG = 1
UGSM_ERR_SIZE_MISMATCH = 2
UGSM_ERR_TOO_SMALL = 3
UGSM_ERR_NO_DEVICE = 4
UGSM_ERR_DEVICE = 5
UGSM_ERR_NOMEM = 6
UGSM_ERR_STATE = 7
UGSM_PENDING = 8   # not an error: not finished yet (ugsm_poll, ugsm_next_done with block = 0)
UGSM_EMPTY = 9     # not an error: nothing outstanding (ugsm_next_done)
UGSM_ERR_PEER = 10  # the fovea shard: another rank failed its part of the step, or none answered within the deadline
UGSM_SHARD_ID_BYTES = 128
UGSM_MAX_LEVELS = 32

# every symbol include/ugsm.h declares (tests check the library exports all of them)
EXPORTS = [
    "ugsm_default_config", "ugsm_abi_version", "ugsm_is_dev_library", "ugsm_status_string", "ugsm_create", "ugsm_destroy",
    "ugsm_last_error", "ugsm_level_dims", "ugsm_level_iterations", "ugsm_level_smooth_passes",
    "ugsm_threshold_schedule", "ugsm_fovea_dims", "ugsm_pixel_iterations", "ugsm_plan_level", "ugsm_match_full", "ugsm_submit_full_host", "ugsm_submit_foveated_host",
    "ugsm_match_foveated", "ugsm_match_foveated_full", "ugsm_submit_full", "ugsm_submit_foveated", "ugsm_submit_full_batch", "ugsm_submit_foveated_batch", "ugsm_submit_full_batch_host", "ugsm_submit_foveated_batch_host",
    "ugsm_wait", "ugsm_wait_all",
    "ugsm_submit_pyramids", "ugsm_submit_fovea_coarse", "ugsm_submit_fovea_fine", "ugsm_triangulate", "ugsm_fovea_mapping", "ugsm_triangulate_fovea", "ugsm_reconstruct_full", "ugsm_stage_pyramid",
    "ugsm_stage_iterate", "ugsm_stage_seed", "ugsm_stage_smooth", "ugsm_stage_weighted_difference", "ugsm_last_iterations", "ugsm_get_kernel_stats",
    "ugsm_stage_lr_check", "ugsm_last_lr_marked", "ugsm_slot_stream",
    "ugsm_reset_kernel_stats", "ugsm_set_profile_events", "ugsm_dev_alloc", "ugsm_dev_free", "ugsm_host_alloc", "ugsm_host_free", "ugsm_copy_to_device", "ugsm_copy_to_host",
    # ABI 5: the queue ...
    "ugsm_poll", "ugsm_enqueue_full", "ugsm_enqueue_foveated", "ugsm_enqueue_full_host", "ugsm_enqueue_foveated_host", "ugsm_enqueue_full_managed",
    "ugsm_enqueue_foveated_managed", "ugsm_flush", "ugsm_next_done", "ugsm_queue_depth", "ugsm_queue_plan",
    # ... and RCCL inside the library
    "ugsm_shard_unique_id", "ugsm_shard_init", "ugsm_shard_init_all", "ugsm_shard_rank", "ugsm_shard_count_ranks", "ugsm_submit_fovea_shard", "ugsm_shard_set_timeout",
    "ugsm_shard_gather", "ugsm_shard_finalize", "ugsm_context_device_bytes",
]
# ... and what include/ugsm_dev.h adds (libugsm_dev.so only)
DEV_EXPORTS = ["ugsm_stage_poly_probe", "ugsm_stage_div3_probe", "ugsm_stage_div_probe"]


class UgsmError(RuntimeError):
    def __init__(self, status: int, what: str, tag=None):
        super().__init__(f"ugsm status {status}: {what}")
        self.status = status
        self.tag = tag   # next_done: the pair whose call failed (the completion HAS been consumed: the caller forgets the tag, then handles the error)


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("levels", C.c_int), ("fovea_levels", C.c_int), ("slots", C.c_int),
                ("kernel_path", C.c_int), ("profile_events", C.c_int), ("march_min_pixels", C.c_int), ("march_np", C.c_int),
                ("march_rows", C.c_int), ("march_smooth", C.c_int), ("early_exit_threshold", C.c_float), ("small_max_pixels", C.c_int),
                ("lr_check_threshold", C.c_float), ("streams", C.c_int), ("batch", C.c_int), ("stream_priority", C.c_int)]


class LevelPlan(C.Structure):
    _fields_ = [("cost_kernel", C.c_int), ("smooth_kernel", C.c_int), ("smooth_rh", C.c_int), ("strip_rows", C.c_int), ("seed_fused", C.c_int),
                ("smooth_tile_rows", C.c_int), ("alone", C.c_int), ("pairs_per_launch", C.c_int)]


class KernelStat(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("level", C.c_int), ("reserved", C.c_int), ("launches", C.c_longlong),
                ("total_ms", C.c_double), ("pixel_launches", C.c_double)]


class Completion(C.Structure):
    _fields_ = [("tag", C.c_uint64), ("status", C.c_int), ("slot", C.c_int), ("call_pairs", C.c_int), ("reserved", C.c_int),
                ("call_index", C.c_longlong), ("done_ns", C.c_longlong), ("result", C.POINTER(C.c_float) * 5)]


_libs = {}


def _elf_dynamic(path: str):
    """(SONAME, [DT_NEEDED ...]) of a 64-bit little-endian ELF shared object, read from its dynamic section; (None, []) if the file is not
    one.  Enough of a reader to compare which HIP runtime ABI two libraries name -- no external tool."""
    import struct
    try:
        with open(path, "rb") as f:
            data = f.read()
        if data[:6] != b"\x7fELF\x02\x01":
            return None, []
        shoff, = struct.unpack_from("<Q", data, 0x28)
        shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
        dyn = next((sec for sec in secs if sec[1] == 6), None)   # SHT_DYNAMIC
        if dyn is None:
            return None, []
        strtab = secs[dyn[6]]                                     # sh_link: its string table
        def name(off):
            a = strtab[4] + off
            return data[a:data.index(b"\0", a)].decode()
        soname, needed = None, []
        for i in range(dyn[5] // 16):
            tag, val = struct.unpack_from("<qQ", data, dyn[4] + 16 * i)
            if tag == 0:
                break
            if tag == 14:
                soname = name(val)
            elif tag == 1:
                needed.append(name(val))
        return soname, needed
    except (OSError, ValueError, struct.error, IndexError, StopIteration):
        return None, []


def _torch_hip_runtime_path():
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    return path if os.path.exists(path) else None


def _share_torch_hip_runtime(lib_path: str = LIB_PATH, torch_hip: str | None = None):
    """PyTorch's ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  A process that loads libugsm.so FIRST (which brings in
    /opt/rocm's runtime) and imports torch afterwards ends up with two HIP runtimes, and the second one finds no GPU ("No HIP GPUs are
    available").  When torch is installed, its runtime is therefore loaded first, globally, and libugsm.so binds to it -- the order
    bench.py and the tools have always used (they import torch first).  A host without torch (the ROS node) is not affected.
    Only when the two name the SAME runtime ABI: libugsm.so was compiled against /opt/rocm's headers and asks for one SONAME
    (DT_NEEDED libamdhip64.so.N); a torch wheel built for another ROCm major carries another, and binding libugsm.so's HIP calls to it
    would be an ABI gamble -- then nothing is preloaded and a warning says so.  UGSM_NO_TORCH_RUNTIME=1 switches the preload off.
    Returns what happened: "preloaded", "off", "no torch", "mismatch" or "failed"."""
    import warnings
    if os.environ.get("UGSM_NO_TORCH_RUNTIME") == "1":
        return "off"
    try:
        path = torch_hip if torch_hip is not None else _torch_hip_runtime_path()
    except Exception as e:  # noqa: BLE001  (a broken torch installation must not keep the library from loading)
        warnings.warn(f"ug_stereomatcher_amd: could not look for PyTorch's HIP runtime ({e}); libugsm.so uses the system's", RuntimeWarning)
        return "failed"
    if path is None:
        return "no torch"
    wanted = next((n for n in _elf_dynamic(lib_path)[1] if n.startswith("libamdhip64.so")), None)
    offered = _elf_dynamic(path)[0]
    if wanted is None or offered != wanted:
        warnings.warn(f"ug_stereomatcher_amd: PyTorch bundles HIP runtime {offered!r} but libugsm.so was built against {wanted!r}: not sharing it -- "
                      "import torch BEFORE this package if both are to see the GPU, or rebuild libugsm.so against torch's ROCm", RuntimeWarning)
        return "mismatch"
    try:
        C.CDLL(path, mode=C.RTLD_GLOBAL)
        return "preloaded"
    except OSError as e:
        warnings.warn(f"ug_stereomatcher_amd: could not preload {path} ({e}); a later `import torch` may not see the GPU", RuntimeWarning)
        return "failed"


def load(dev: bool = False):
    """Loads libugsm.so (dev: libugsm_dev.so); raises if it has not been built (python __graft_entry__.py / make -C csrc)."""
    if dev in _libs:
        return _libs[dev]
    path = DEV_LIB_PATH if dev else LIB_PATH
    if not os.path.exists(path):
        raise UgsmError(UGSM_ERR_NO_DEVICE, f"{path} not built: run `make -C ug_stereomatcher_amd/csrc`")
    if not _libs:
        _share_torch_hip_runtime(path)
    lib = C.CDLL(path)
    vp, ip, fp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)
    i = C.c_int
    lib.ugsm_default_config.argtypes = [C.POINTER(Config)]
    lib.ugsm_default_config.restype = None
    lib.ugsm_abi_version.restype = i
    lib.ugsm_is_dev_library.restype = i
    lib.ugsm_status_string.argtypes = [i]
    lib.ugsm_status_string.restype = C.c_char_p
    lib.ugsm_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.ugsm_destroy.argtypes = [vp]
    lib.ugsm_destroy.restype = None
    lib.ugsm_last_error.argtypes = [vp]
    lib.ugsm_last_error.restype = C.c_char_p
    lib.ugsm_level_dims.argtypes = [i, i, i, ip, ip]
    lib.ugsm_level_iterations.argtypes = [i]
    lib.ugsm_level_smooth_passes.argtypes = [i]
    lib.ugsm_threshold_schedule.argtypes = [i, fp]
    lib.ugsm_fovea_dims.argtypes = [i, i, i, i, ip, ip]
    lib.ugsm_pixel_iterations.argtypes = [i, i, i, i]
    lib.ugsm_pixel_iterations.restype = C.c_longlong
    lib.ugsm_plan_level.argtypes = [C.POINTER(Config), i, i, i, C.POINTER(LevelPlan)]
    lib.ugsm_match_full.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp]
    lib.ugsm_submit_full_host.argtypes = [vp, i, vp, vp, i, i, i, vp, vp, vp]
    lib.ugsm_submit_foveated_host.argtypes = [vp, i, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp]
    lib.ugsm_match_foveated.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp]
    lib.ugsm_match_foveated_full.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp]
    lib.ugsm_submit_full.argtypes = [vp, i, vp, vp, i, i, i, vp]
    lib.ugsm_submit_foveated.argtypes = [vp, i, vp, vp, i, i, i, i, i, vp, vp, vp]
    pp = C.POINTER(vp)
    lib.ugsm_submit_full_batch.argtypes = [vp, i, i, pp, pp, i, i, i, pp]
    lib.ugsm_submit_foveated_batch.argtypes = [vp, i, i, pp, pp, i, i, i, ip, ip, pp, pp, pp]
    lib.ugsm_submit_full_batch_host.argtypes = [vp, i, i, pp, pp, i, i, i, pp, pp, pp]
    lib.ugsm_submit_foveated_batch_host.argtypes = [vp, i, i, pp, pp, i, i, i, ip, ip, pp, pp, pp]
    lib.ugsm_wait.argtypes = [vp, i]
    lib.ugsm_wait_all.argtypes = [vp]
    lib.ugsm_submit_pyramids.argtypes = [vp, i, vp, vp, i, i, i]
    lib.ugsm_submit_fovea_coarse.argtypes = [vp, i, vp]
    lib.ugsm_submit_fovea_fine.argtypes = [vp, i, vp, i, i, vp]
    lib.ugsm_triangulate.argtypes = [vp, i, vp, vp, i, i, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    lib.ugsm_fovea_mapping.argtypes = [i, i, i, i, C.POINTER(i), C.POINTER(i), C.POINTER(C.c_float)]
    lib.ugsm_triangulate_fovea.argtypes = [vp, i, vp, vp, i, i, i, i, i, C.c_float, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    lib.ugsm_reconstruct_full.argtypes = [vp, i, vp, vp, vp, i, i, i, i, vp]
    lib.ugsm_stage_pyramid.argtypes = [vp, vp, i, i, i, i, vp]
    lib.ugsm_stage_iterate.argtypes = [vp, vp, vp, vp, i, i, i, i, i, i, i, vp]
    lib.ugsm_stage_seed.argtypes = [vp, vp, i, i, vp, i, i, i, i, i, i]
    lib.ugsm_stage_smooth.argtypes = [vp, vp, i, i, i, i]
    if dev:
        lib.ugsm_stage_poly_probe.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i]
        lib.ugsm_stage_div3_probe.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i]
        lib.ugsm_stage_div_probe.argtypes = [vp, vp, vp, vp, i]
    lib.ugsm_stage_weighted_difference.argtypes = [vp, vp, vp, i, i, C.POINTER(C.c_float)]
    lib.ugsm_last_iterations.argtypes = [vp, i, C.POINTER(i)]
    lib.ugsm_stage_lr_check.argtypes = [vp, vp, vp, i, i, C.c_float, C.POINTER(C.c_longlong)]
    lib.ugsm_last_lr_marked.argtypes = [vp, i]
    lib.ugsm_last_lr_marked.restype = C.c_longlong
    lib.ugsm_slot_stream.argtypes = [vp, i, C.POINTER(vp)]
    lib.ugsm_get_kernel_stats.argtypes = [vp, C.POINTER(KernelStat), i]
    lib.ugsm_reset_kernel_stats.argtypes = [vp]
    lib.ugsm_set_profile_events.argtypes = [vp, i]
    lib.ugsm_dev_alloc.argtypes = [vp, C.POINTER(vp), C.c_longlong]
    lib.ugsm_dev_free.argtypes = [vp, vp]
    lib.ugsm_host_alloc.argtypes = [vp, C.POINTER(vp), C.c_longlong]
    lib.ugsm_host_free.argtypes = [vp, vp]
    lib.ugsm_copy_to_device.argtypes = [vp, vp, vp, C.c_longlong]
    lib.ugsm_copy_to_host.argtypes = [vp, vp, vp, C.c_longlong]
    u64 = C.c_uint64
    lib.ugsm_poll.argtypes = [vp, i]
    lib.ugsm_context_device_bytes.argtypes = [vp]
    lib.ugsm_context_device_bytes.restype = C.c_longlong
    lib.ugsm_enqueue_full.argtypes = [vp, vp, vp, i, i, i, vp, u64]
    lib.ugsm_enqueue_foveated.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, u64]
    lib.ugsm_enqueue_full_host.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, u64]
    lib.ugsm_enqueue_foveated_host.argtypes = [vp, vp, vp, i, i, i, i, i, vp, vp, vp, vp, vp, u64]
    lib.ugsm_enqueue_full_managed.argtypes = [vp, vp, vp, i, i, i, u64]
    lib.ugsm_enqueue_foveated_managed.argtypes = [vp, vp, vp, i, i, i, i, i, i, u64]
    lib.ugsm_flush.argtypes = [vp]
    lib.ugsm_next_done.argtypes = [vp, C.POINTER(Completion), i]
    lib.ugsm_queue_depth.argtypes = [vp, ip, ip, ip]
    lib.ugsm_queue_plan.argtypes = [C.POINTER(Config), i, ip, i]
    lib.ugsm_shard_unique_id.argtypes = [vp]
    lib.ugsm_shard_init.argtypes = [vp, vp, i, i]
    lib.ugsm_shard_init_all.argtypes = [C.POINTER(vp), i]
    lib.ugsm_shard_rank.argtypes = [vp, ip, ip]
    lib.ugsm_shard_count_ranks.argtypes = [vp, ip]
    lib.ugsm_submit_fovea_shard.argtypes = [vp, i, vp, vp, i, i, i, i, i, vp, i]
    lib.ugsm_shard_set_timeout.argtypes = [vp, C.c_longlong]
    lib.ugsm_shard_gather.argtypes = [vp, i, vp, C.c_longlong, vp, i]
    lib.ugsm_shard_finalize.argtypes = [vp]
    if bool(lib.ugsm_is_dev_library()) != bool(dev):
        raise UgsmError(UGSM_ERR_STATE, f"{path} is not the {'development' if dev else 'product'} build")
    _libs[dev] = lib
    return lib


def status_string(st: int) -> str:
    return load().ugsm_status_string(st).decode()


# ---- pure-host geometry (no GPU needed) -------------------------------------------------

def level_dims(W: int, H: int, levels: int = 14):
    w = (C.c_int * UGSM_MAX_LEVELS)()
    h = (C.c_int * UGSM_MAX_LEVELS)()
    st = load().ugsm_level_dims(W, H, levels, w, h)
    if st:
        raise UgsmError(st, status_string(st))
    return list(w[:levels]), list(h[:levels])


def level_iterations(level: int) -> int:
    return load().ugsm_level_iterations(level)


def level_smooth_passes(level: int) -> int:
    return load().ugsm_level_smooth_passes(level)


def threshold_schedule(mi: int) -> np.ndarray:
    out = np.zeros(mi, np.float32)
    st = load().ugsm_threshold_schedule(mi, out.ctypes.data_as(C.POINTER(C.c_float)))
    if st:
        raise UgsmError(st, status_string(st))
    return out


def fovea_dims(W: int, H: int, levels: int = 14, fovea_levels: int = 7):
    fw, fh = C.c_int(), C.c_int()
    st = load().ugsm_fovea_dims(W, H, levels, fovea_levels, C.byref(fw), C.byref(fh))
    if st:
        raise UgsmError(st, status_string(st))
    return fw.value, fh.value


def pixel_iterations(W: int, H: int, levels: int = 14, fovea_levels: int = 0) -> int:
    return int(load().ugsm_pixel_iterations(W, H, levels, fovea_levels))


# ---- context -----------------------------------------------------------------------------

class Context:
    """Owns a ugsm_ctx*.  One per device; calls must be serialised by the caller."""

    def __init__(self, device: int = 0, levels: int = 14, fovea_levels: int = 7, slots: int = 1,
                 kernel_path: int = 0, profile_events: int = 0, march_min_pixels: int = 0, march_np: int = 0,
                 march_rows: int = 0, march_smooth: int = 0, early_exit_threshold: float = 0.0, small_max_pixels: int = 0,
                 lr_check_threshold: float = 0.0, streams: int = 0, batch: int = 0, stream_priority: int = 0, dev: bool | None = None):
        # libugsm_dev.so when asked for, or when the configuration needs a kernel only it has
        self.dev = bool(dev) if dev is not None else (kernel_path == 1 or march_min_pixels < 0)
        lib = load(self.dev)
        cfg = Config()
        lib.ugsm_default_config(C.byref(cfg))
        # full-mode users never name fovea_levels; keep the default legal for short pyramids
        cfg.device, cfg.levels, cfg.fovea_levels, cfg.slots = device, levels, min(fovea_levels, levels), slots
        cfg.kernel_path, cfg.profile_events = kernel_path, int(profile_events)
        cfg.march_min_pixels, cfg.march_np, cfg.march_rows = int(march_min_pixels), int(march_np), int(march_rows)
        cfg.march_smooth = int(march_smooth)
        cfg.early_exit_threshold = float(early_exit_threshold)
        cfg.small_max_pixels = int(small_max_pixels)
        cfg.lr_check_threshold = float(lr_check_threshold)
        cfg.streams = int(streams)
        cfg.batch, cfg.stream_priority = int(batch), int(stream_priority)
        self.cfg = cfg
        self._pinned = []
        self._h = C.c_void_p()
        st = lib.ugsm_create(C.byref(cfg), C.byref(self._h))
        if st:
            self._h = None
            raise UgsmError(st, status_string(st) + " (libugsm needs a HIP device; there is no CPU fallback)")
        self.lib = lib

    def close(self):
        if getattr(self, "_h", None):
            for p in getattr(self, "_pinned", []):
                self.lib.ugsm_host_free(self._h, p)
            self._pinned = []
            self.lib.ugsm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def check(self, st: int):
        if st:
            raise UgsmError(st, f"{status_string(st)}: {self.lib.ugsm_last_error(self._h).decode()}")

    @property
    def handle(self):
        return self._h

    # device memory through the C-ABI (tests / C hosts); bench.py uses torch tensors instead
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.lib.ugsm_dev_alloc(self._h, C.byref(p), nbytes))
        return p.value

    def free(self, ptr: int):
        self.check(self.lib.ugsm_dev_free(self._h, ptr))

    def to_device(self, arr: np.ndarray) -> int:
        arr = np.ascontiguousarray(arr)
        p = self.alloc(arr.nbytes)
        self.check(self.lib.ugsm_copy_to_device(self._h, p, arr.ctypes.data, arr.nbytes))
        return p

    def to_host(self, ptr: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype)
        self.check(self.lib.ugsm_copy_to_host(self._h, out.ctypes.data, ptr, out.nbytes))
        return out

    def host_array(self, shape, dtype=np.float32) -> np.ndarray:
        """A numpy array in page-locked host memory (ugsm_host_alloc); freed when the context closes."""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.check(self.lib.ugsm_host_alloc(self._h, C.byref(p), nbytes))
        self._pinned.append(p.value)
        buf = (C.c_char * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def triangulate(self, d_dispx: int, d_dispy: int, W: int, H: int, P1, P2, d_xyz: int, slot: int = 0):
        """SURVEY 8f row f-1 (getPointCloud.cpp:886-949): X, Y, Z planes from device (dx, dy) planes."""
        p1 = np.ascontiguousarray(P1, np.float64).reshape(12)
        p2 = np.ascontiguousarray(P2, np.float64).reshape(12)
        dp = C.POINTER(C.c_double)
        self.check(self.lib.ugsm_triangulate(self._h, slot, d_dispx, d_dispy, W, H, p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), d_xyz))
        self.check(self.lib.ugsm_wait(self._h, slot))

    def triangulate_fovea(self, d_stackx: int, d_stacky: int, fovW: int, fovH: int, src_level: int, left: int, upper: int, scale,
                          P1, P2, d_xyz: int, slot: int = 0):
        """Row f-1, foveated branch (getPointCloud.cpp:892-903): X, Y, Z planes for one level of the fovea stacks."""
        p1 = np.ascontiguousarray(P1, np.float64).reshape(12)
        p2 = np.ascontiguousarray(P2, np.float64).reshape(12)
        dp = C.POINTER(C.c_double)
        self.check(self.lib.ugsm_triangulate_fovea(self._h, slot, d_stackx, d_stacky, fovW, fovH, src_level, left, upper,
                                                   C.c_float(float(scale)), p1.ctypes.data_as(dp), p2.ctypes.data_as(dp), d_xyz))
        self.check(self.lib.ugsm_wait(self._h, slot))

    def reconstruct_full(self, d_stackH: int, d_stackV: int, d_stackC: int, W: int, H: int, d_out3: int, off_x: int = 0, off_y: int = 0,
                         slot: int = 0):
        """Row f-3 (MatchGPULib.cpp:2589-2701): full-resolution field from the foveated stacks."""
        self.check(self.lib.ugsm_reconstruct_full(self._h, slot, d_stackH, d_stackV, d_stackC, W, H, off_x, off_y, d_out3))
        self.check(self.lib.ugsm_wait(self._h, slot))

    # ---- B pairs per call (ugsm_submit_*_batch): lists of device pointers ---------------------------------------------
    @staticmethod
    def _ptrs(ps):
        return (C.c_void_p * len(ps))(*[int(p) if p is not None else None for p in ps])

    def submit_full_batch(self, slot: int, d_rgbL, d_rgbR, W: int, H: int, stride: int, d_out):
        n = len(d_rgbL)
        self.check(self.lib.ugsm_submit_full_batch(self._h, slot, n, self._ptrs(d_rgbL), self._ptrs(d_rgbR), W, H, stride, self._ptrs(d_out)))

    def submit_foveated_batch(self, slot: int, d_rgbL, d_rgbR, W: int, H: int, stride: int, offsets, d_stack, d_pyrL=None, d_pyrR=None):
        n = len(d_rgbL)
        ox = (C.c_int * n)(*[int(o[0]) for o in offsets]) if offsets is not None else None
        oy = (C.c_int * n)(*[int(o[1]) for o in offsets]) if offsets is not None else None
        self.check(self.lib.ugsm_submit_foveated_batch(self._h, slot, n, self._ptrs(d_rgbL), self._ptrs(d_rgbR), W, H, stride, ox, oy, self._ptrs(d_stack),
                                                       self._ptrs(d_pyrL) if d_pyrL is not None else None,
                                                       self._ptrs(d_pyrR) if d_pyrR is not None else None))

    def submit_full_batch_host(self, slot: int, rgbL, rgbR, W: int, H: int, stride: int, outs):
        """rgbL / rgbR: lists of page-locked uint8 arrays (host_array); outs: list of page-locked (3, H, W) float32 arrays."""
        n = len(rgbL)
        self.check(self.lib.ugsm_submit_full_batch_host(self._h, slot, n, self._ptrs([a.ctypes.data for a in rgbL]), self._ptrs([a.ctypes.data for a in rgbR]),
                                                        W, H, stride, self._ptrs([o[0].ctypes.data for o in outs]), self._ptrs([o[1].ctypes.data for o in outs]),
                                                        self._ptrs([o[2].ctypes.data for o in outs])))

    def submit_foveated_batch_host(self, slot: int, rgbL, rgbR, W: int, H: int, stride: int, offsets, stacks):
        """stacks: list of page-locked (3, F, fovH, fovW) float32 arrays."""
        n = len(rgbL)
        ox = (C.c_int * n)(*[int(o[0]) for o in offsets]) if offsets is not None else None
        oy = (C.c_int * n)(*[int(o[1]) for o in offsets]) if offsets is not None else None
        self.check(self.lib.ugsm_submit_foveated_batch_host(self._h, slot, n, self._ptrs([a.ctypes.data for a in rgbL]), self._ptrs([a.ctypes.data for a in rgbR]),
                                                            W, H, stride, ox, oy, self._ptrs([t[0].ctypes.data for t in stacks]),
                                                            self._ptrs([t[1].ctypes.data for t in stacks]), self._ptrs([t[2].ctypes.data for t in stacks])))

    # ---- the queue (ugsm_enqueue_* / ugsm_flush / ugsm_next_done): the library owns the slots ----------------------------------------
    def enqueue_full(self, d_rgbL: int, d_rgbR: int, W: int, H: int, stride: int, d_out: int, tag: int):
        self.check(self.lib.ugsm_enqueue_full(self._h, d_rgbL, d_rgbR, W, H, stride, d_out, tag))

    def enqueue_foveated(self, d_rgbL: int, d_rgbR: int, W: int, H: int, stride: int, off, d_stack: int, tag: int, d_pyrL=None, d_pyrR=None):
        self.check(self.lib.ugsm_enqueue_foveated(self._h, d_rgbL, d_rgbR, W, H, stride, int(off[0]), int(off[1]), d_stack, d_pyrL, d_pyrR, tag))

    def enqueue_full_host(self, rgbL: np.ndarray, rgbR: np.ndarray, out3: np.ndarray, tag: int):
        """rgbL / rgbR: page-locked (H, W, 3) uint8 (host_array); out3: page-locked (3, H, W) float32."""
        H, W = rgbL.shape[:2]
        self.check(self.lib.ugsm_enqueue_full_host(self._h, rgbL.ctypes.data, rgbR.ctypes.data, W, H, rgbL.strides[0], out3[0].ctypes.data,
                                                   out3[1].ctypes.data, out3[2].ctypes.data, tag))

    def enqueue_foveated_host(self, rgbL: np.ndarray, rgbR: np.ndarray, off, stack3: np.ndarray, tag: int, pyrL=None, pyrR=None):
        H, W = rgbL.shape[:2]
        self.check(self.lib.ugsm_enqueue_foveated_host(self._h, rgbL.ctypes.data, rgbR.ctypes.data, W, H, rgbL.strides[0], int(off[0]), int(off[1]),
                                                       stack3[0].ctypes.data, stack3[1].ctypes.data, stack3[2].ctypes.data,
                                                       pyrL.ctypes.data if pyrL is not None else None, pyrR.ctypes.data if pyrR is not None else None, tag))

    def enqueue_full_managed(self, rgbL: np.ndarray, rgbR: np.ndarray, tag: int):
        """Any host memory: the images are copied before the call returns; the results come back in ugsm_completion.result."""
        H, W = rgbL.shape[:2]
        self.check(self.lib.ugsm_enqueue_full_managed(self._h, rgbL.ctypes.data, rgbR.ctypes.data, W, H, rgbL.strides[0], tag))

    def enqueue_foveated_managed(self, rgbL: np.ndarray, rgbR: np.ndarray, off, want_pyramids: bool, tag: int):
        H, W = rgbL.shape[:2]
        self.check(self.lib.ugsm_enqueue_foveated_managed(self._h, rgbL.ctypes.data, rgbR.ctypes.data, W, H, rgbL.strides[0], int(off[0]), int(off[1]),
                                                          1 if want_pyramids else 0, tag))

    def flush(self):
        self.check(self.lib.ugsm_flush(self._h))

    def next_done(self, block: bool = True):
        """The oldest pair not yet reported as a Completion, or None (block=False: not finished yet, or nothing outstanding; block=True:
        nothing outstanding).  A completion whose call failed raises UgsmError with `.tag` set: the pair has been reported -- the library will not
        name it again -- so whoever keeps per-tag state drops it before handling the error."""
        c = Completion()
        st = self.lib.ugsm_next_done(self._h, C.byref(c), 1 if block else 0)
        if st in (UGSM_PENDING, UGSM_EMPTY):
            return None
        self.check(st)
        if c.status:
            raise UgsmError(c.status, f"pair {c.tag}: {status_string(c.status)}: {self.lib.ugsm_last_error(self._h).decode()}", tag=int(c.tag))
        return c

    def drain(self):
        """Flushes and fetches every outstanding completion, oldest first."""
        out = []
        while True:
            c = self.next_done(True)
            if c is None:
                return out
            out.append(c)

    def queue_depth(self):
        w, f, u = C.c_int(), C.c_int(), C.c_int()
        self.check(self.lib.ugsm_queue_depth(self._h, C.byref(w), C.byref(f), C.byref(u)))
        return w.value, f.value, u.value

    @staticmethod
    def managed_planes(c: "Completion", shapes):
        """numpy views (valid until the next next_done on the context) of a managed completion's result planes; shapes: one per plane wanted."""
        out = []
        for k, shp in enumerate(shapes):
            n = int(np.prod(shp))
            out.append(np.ctypeslib.as_array(c.result[k], shape=(n,)).reshape(shp))
        return out

    # ---- the fovea shard with RCCL inside the library (ugsm_shard_*) ------------------------------------------------------------------
    def shard_init(self, id128: bytes, rank: int, world: int):
        buf = (C.c_char * UGSM_SHARD_ID_BYTES).from_buffer_copy(id128)
        self.check(self.lib.ugsm_shard_init(self._h, buf, rank, world))

    def shard_count_ranks(self) -> int:
        n = C.c_int()
        self.check(self.lib.ugsm_shard_count_ranks(self._h, C.byref(n)))
        return n.value

    def submit_fovea_shard(self, slot: int, d_rgbL: int, d_rgbR: int, W: int, H: int, stride: int, off, d_stack: int, src_rank: int = 0):
        self.check(self.lib.ugsm_submit_fovea_shard(self._h, slot, d_rgbL, d_rgbR, W, H, stride, int(off[0]), int(off[1]), d_stack, src_rank))

    def shard_set_timeout(self, milliseconds: int):
        self.check(self.lib.ugsm_shard_set_timeout(self._h, int(milliseconds)))

    def shard_init_all(self, others=()):
        """One process, several GPUs: this context and `others` (each on its own device) become ranks 0 .. n-1 of one communicator."""
        hs = (C.c_void_p * (1 + len(others)))(self._h, *[o._h for o in others])
        self.check(self.lib.ugsm_shard_init_all(hs, 1 + len(others)))

    def shard_gather(self, slot: int, d_stack: int, stack_floats: int, d_all, dst_rank: int = 0):
        self.check(self.lib.ugsm_shard_gather(self._h, slot, d_stack, stack_floats, d_all, dst_rank))

    def shard_finalize(self):
        self.check(self.lib.ugsm_shard_finalize(self._h))

    def device_bytes(self) -> int:
        return int(self.lib.ugsm_context_device_bytes(self._h))

    def kernel_stats(self):
        """One dict per (kernel, pyramid level) with harvested launches; level -1 = not tied to a level."""
        cap = 512
        arr = (KernelStat * cap)()
        n = self.lib.ugsm_get_kernel_stats(self._h, arr, cap)
        return [dict(name=arr[k].name.decode(), level=int(arr[k].level), launches=int(arr[k].launches), total_ms=float(arr[k].total_ms),
                     pixel_launches=float(arr[k].pixel_launches)) for k in range(min(n, cap))]

    def set_profile_events(self, mode: int):
        self.check(self.lib.ugsm_set_profile_events(self._h, int(mode)))

    def reset_kernel_stats(self):
        self.check(self.lib.ugsm_reset_kernel_stats(self._h))


def plan_level(W: int, H: int, alone: bool = True, **cfg_fields):
    """Which kernels a W x H level runs (host only): dict of ugsm_level_plan; cfg_fields override the default ugsm_config.
    alone: the call has the chip to itself (what the library decides per call from what is in flight; include/ugsm.h)."""
    lib = load(bool(cfg_fields.get("kernel_path") == 1 or cfg_fields.get("march_min_pixels", 0) < 0 or cfg_fields.pop("dev", False)))
    cfg = Config()
    lib.ugsm_default_config(C.byref(cfg))
    for k, v in cfg_fields.items():
        setattr(cfg, k, v)
    out = LevelPlan()
    st = lib.ugsm_plan_level(C.byref(cfg), 1 if alone else 0, W, H, C.byref(out))
    if st != 0:
        raise UgsmError(st, "ugsm_plan_level")
    return {k: int(getattr(out, k)) for k in ("cost_kernel", "smooth_kernel", "smooth_rh", "strip_rows", "seed_fused", "smooth_tile_rows", "alone", "pairs_per_launch")}


def shard_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 makes one and hands it to the other ranks out of band)."""
    buf = (C.c_char * UGSM_SHARD_ID_BYTES)()
    st = load().ugsm_shard_unique_id(buf)
    if st != 0:
        raise UgsmError(st, "ugsm_shard_unique_id: " + status_string(st))
    return bytes(buf)


def queue_plan(n_pairs: int, **cfg_fields):
    """The calls the queue forms from a burst of n_pairs pairs enqueued from idle and flushed (host only)."""
    lib = load()
    cfg = Config()
    lib.ugsm_default_config(C.byref(cfg))
    for k, v in cfg_fields.items():
        setattr(cfg, k, v)
    cap = max(1, n_pairs)
    sizes = (C.c_int * cap)()
    n = lib.ugsm_queue_plan(C.byref(cfg), n_pairs, sizes, cap)
    if n < 0:
        raise UgsmError(UGSM_ERR_BAD_ARG, "ugsm_queue_plan")
    return list(sizes[:n])


def fovea_mapping(W: int, H: int, src_level: int, dest_level: int = 0):
    """getPointCloud.cpp:387-484 -> (left_margin, upper_margin, scale); host only."""
    lib = load()
    l, u, sc = C.c_int(), C.c_int(), C.c_float()
    st = lib.ugsm_fovea_mapping(W, H, src_level, dest_level, C.byref(l), C.byref(u), C.byref(sc))
    if st != 0:
        raise UgsmError(st, "ugsm_fovea_mapping")
    return l.value, u.value, np.float32(sc.value)
