"""ctypes mirror of include/samtools_amd.h (structures, constants, prototypes)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "lib", "libsamtools_amd.so")

if not os.path.exists(_LIB_PATH):
    raise ImportError(
        "samtools_amd: %s is missing -- build it with `make -C samtools_amd/csrc` "
        "(or __graft_entry__.build()); the engine has no Python/CPU fallback" % _LIB_PATH)

lib = C.CDLL(_LIB_PATH, mode=C.RTLD_GLOBAL)

STA_OK = 0
STA_ERR_NO_DEVICE = -2
STA_ERR_IO = -7
STA_MEM_HOST = 0
STA_MEM_DEVICE = 1


class MPLP:
    NO_ORPHAN = 1 << 3
    REALN = 1 << 4
    REDO_BAQ = 1 << 6
    ILLUMINA13 = 1 << 7
    SMART_OVERLAPS = 1 << 10
    PRINT_MAPQ_CHAR = 1 << 11
    PRINT_QPOS = 1 << 12
    PRINT_QNAME = 1 << 13
    PRINT_FLAG = 1 << 14
    PRINT_RNAME = 1 << 15
    PRINT_POS = 1 << 16
    PRINT_MAPQ = 1 << 17
    PRINT_PNEXT = 1 << 20
    PRINT_RLEN = 1 << 24
    PRINT_QPOS5 = 1 << 26
    DEFAULT = NO_ORPHAN | REALN | SMART_OVERLAPS


class Reads(C.Structure):
    """sta_reads: pointers are raw addresses (host or device, see Window.mem)."""
    _fields_ = [
        ("n_reads", C.c_int64),
        ("pos", C.c_void_p), ("flag", C.c_void_p), ("mapq", C.c_void_p), ("aux", C.c_void_p),
        ("l_qseq", C.c_void_p), ("cig_off", C.c_void_p), ("base_off8", C.c_void_p), ("mtid", C.c_void_p),
        ("mpos", C.c_void_p), ("isize", C.c_void_p), ("name_off", C.c_void_p), ("cigar", C.c_void_p),
        ("seq", C.c_void_p), ("qual", C.c_void_p), ("bq", C.c_void_p), ("names", C.c_void_p),
        ("n_cigar_total", C.c_uint64), ("n_bases_total", C.c_uint64), ("n_name_bytes", C.c_uint64),
        ("n_xcols", C.c_int32), ("xcol_off", C.c_void_p), ("xcol_text", C.c_void_p), ("n_xcol_bytes", C.c_uint64),
        ("mod_off", C.c_void_p), ("mod_qpos", C.c_void_p), ("mod_toff", C.c_void_p), ("mod_text", C.c_void_p),
        ("n_mod_entries", C.c_uint64), ("n_mod_bytes", C.c_uint64),
        # device staging out of raw BAM records (host drivers only; zero here: everything staged by the caller)
        ("raw_first", C.c_int64), ("n_raw_pieces", C.c_int32), ("raw_pieces", C.c_void_p), ("raw_rec_off", C.c_void_p), ("raw_verify", C.c_int32),
        # what every read found in the reference's name hash, worked out by the caller in file order (NULL: the engine replays the hash from the staged names)
        ("olap_clip", C.c_void_p), ("olap_mate", C.c_void_p),
    ]


class Window(C.Structure):
    _fields_ = [
        ("tid", C.c_int32), ("origin", C.c_int64), ("col_beg", C.c_int32), ("col_end", C.c_int32),
        ("tname", C.c_char_p), ("tlen", C.c_int64),
        ("n_files", C.c_int32), ("files", C.POINTER(Reads)), ("mem", C.c_int32),
        ("has_bed", C.c_int32), ("n_bed", C.c_int64), ("bed_beg", C.c_void_p), ("bed_end", C.c_void_p),
        ("has_reg", C.c_int32), ("reg_beg", C.c_int64), ("reg_end", C.c_int64),
    ]


class MplpParams(C.Structure):
    _fields_ = [
        ("min_mq", C.c_int32), ("min_baseQ", C.c_int32), ("capQ_thres", C.c_int32), ("max_depth", C.c_int32),
        ("all", C.c_int32), ("rev_del", C.c_int32), ("rflag_require", C.c_int32), ("rflag_filter", C.c_int32),
        ("flag", C.c_int32), ("no_ins", C.c_int32), ("no_del", C.c_int32), ("no_ends", C.c_int32),
        ("has_fai", C.c_int32), ("n_tags", C.c_int32), ("tag_sep", C.c_int32), ("min_qlen", C.c_int32), ("no_ins_mods", C.c_int32),
    ]

    @classmethod
    def defaults(cls):
        """bam_mpileup() defaults (bam_plcmd.c:1083-1094)."""
        return cls(min_mq=0, min_baseQ=13, capQ_thres=0, max_depth=8000, all=0, rev_del=0, rflag_require=0,
                   rflag_filter=4 | 256 | 512 | 1024, flag=MPLP.DEFAULT, no_ins=0, no_del=0, no_ends=0, has_fai=0, n_tags=0, tag_sep=ord(','), min_qlen=0, no_ins_mods=0)


class DepthParams(C.Structure):
    _fields_ = [
        ("flag", C.c_int32), ("incl_flag", C.c_int32), ("require_flag", C.c_int32),
        ("min_qual", C.c_int32), ("min_mqual", C.c_int32), ("min_len", C.c_int32),
        ("skip_del", C.c_int32), ("all_pos", C.c_int32), ("remove_overlaps", C.c_int32),
    ]

    @classmethod
    def defaults(cls):
        """main_depth() defaults (bam2depth.c:740-754)."""
        return cls(flag=4 | 256 | 1024 | 512, incl_flag=0, require_flag=0, min_qual=0, min_mqual=0, min_len=0,
                   skip_del=1, all_pos=0, remove_overlaps=0)


class PlanInfo(C.Structure):
    _fields_ = [("out_bytes", C.c_uint64), ("n_lines", C.c_uint64), ("n_data_cols", C.c_uint64),
                ("n_kept_reads", C.c_uint64), ("piled_bases", C.c_uint64), ("n_maxcnt_dropped", C.c_uint64)]


class CalmdParams(C.Structure):
    _fields_ = [("flag", C.c_int32), ("max_nm", C.c_int32), ("capQ", C.c_int32)]


class GlfParams(C.Structure):
    _fields_ = [("min_baseQ", C.c_int32), ("max_depth", C.c_int32), ("theta", C.c_double)]


class GlfCol(C.Structure):
    _fields_ = [("n_plp", C.c_int32), ("n", C.c_int32), ("flags", C.c_int32), ("qsum", C.c_float * 4), ("p", C.c_float * 25)]


class ConsParams(C.Structure):
    """sta_cons_params: consensus_opts of `samtools consensus` (bam_consensus.c:211-260)"""
    _fields_ = [(n, C.c_int32) for n in ("mode use_qual min_qual adj_qual use_mqual nm_adjust nm_halo sc_cost low_mqual high_mqual min_depth "
                                          "cons_cutoff ambig default_qual excl_flags incl_flags min_mqual want_pileup").split()] \
        + [(n, C.c_double) for n in "scale_mqual call_fract het_fract P_het P_indel het_scale homopoly_fix homopoly_redux".split()] \
        + [("qcal", (C.c_int32 * 101) * 3)]

    @classmethod
    def defaults(cls, **kw):
        """main_consensus's defaults (bam_consensus.c:3152-3194); keyword arguments override fields"""
        p = cls(mode=2, adj_qual=1, use_mqual=1, scale_mqual=1.0, nm_adjust=1, nm_halo=50, sc_cost=60, low_mqual=1, high_mqual=60, min_depth=1,
                call_fract=0.75, het_fract=0.5, cons_cutoff=10, default_qual=10, excl_flags=4 | 256 | 512 | 1024, P_het=1e-3, P_indel=2e-4,
                het_scale=1.0, homopoly_redux=0.01)
        for k in range(3):
            for i in range(101):
                p.qcal[k][i] = i
        for k, v in kw.items():
            setattr(p, k, v)
        return p


class ConsCol(C.Structure):
    _fields_ = [("depth", C.c_int32), ("base", C.c_int32), ("qual", C.c_int32)]


class ConsInfo(C.Structure):
    _fields_ = [("n_cols", C.c_uint64), ("n_entries", C.c_uint64), ("n_kept_reads", C.c_uint64)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double)]


_P = C.c_void_p
_PROTOS = {
    "sta_engine_create": (C.c_int, [C.POINTER(_P), C.c_int, _P]),
    "sta_engine_destroy": (None, [_P]),
    "sta_last_error": (C.c_char_p, [_P]),
    "sta_device_count": (C.c_int, []),
    "sta_version": (C.c_char_p, []),
    "sta_set_reference": (C.c_int, [_P, C.c_int32, _P, C.c_int64, C.c_int32]),
    "sta_clear_references": (None, [_P]),
    "sta_stage_window": (C.c_int, [_P, C.POINTER(Window)]),
    "sta_mpileup_plan": (C.c_int, [_P, C.POINTER(MplpParams), C.POINTER(PlanInfo)]),
    "sta_mpileup_emit": (C.c_int, [_P, _P, C.c_uint64]),
    "sta_mpileup_run": (C.c_int, [_P, C.POINTER(MplpParams), _P, C.c_uint64, C.POINTER(PlanInfo)]),
    "sta_depth_plan": (C.c_int, [_P, C.POINTER(DepthParams), C.POINTER(PlanInfo)]),
    "sta_depth_emit": (C.c_int, [_P, _P, C.c_uint64]),
    "sta_depth_run": (C.c_int, [_P, C.POINTER(DepthParams), _P, C.c_uint64, C.POINTER(PlanInfo)]),
    "sta_depth_counts_dev": (_P, [_P]),
    "sta_fetch_output": (C.c_int, [_P, _P, C.c_uint64]),
    "sta_fetch_output_at": (C.c_int, [_P, _P, C.c_uint64, C.c_uint64]),
    "sta_sync": (C.c_int, [_P]),
    "sta_profile_enable": (None, [_P, C.c_int]),
    "sta_profile_only": (None, [_P, C.c_char_p]),
    "sta_profile_reset": (None, [_P]),
    "sta_profile_get": (C.c_int, [_P, C.POINTER(KernelTime), C.c_int]),
    "sta_plp_plan": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(PlanInfo)]),
    "sta_plp_emit": (C.c_int, [_P, _P, C.c_uint64]),
    "sta_fetch_col_offsets": (C.c_int, [_P, _P, C.c_uint64]),
    "sta_stage_raw_reads": (C.c_uint64, [_P]),
    "sta_bgzf_inflate": (C.c_int, [C.c_int32, _P, C.c_uint64, _P, C.c_int32, _P, C.c_uint64, _P, _P]),
    "sta_fetch_read_state": (C.c_int, [_P, C.c_int32, _P, _P]),
    "sta_main_mpileup": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "sta_main_depth": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "sta_main_capture": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "sta_capture_free": (None, [_P]),
    "sta_main_capture_device": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_uint64), C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "sta_capture_device_take": (C.c_int, [_P, C.c_uint64]),
    "sta_glf_plan": (C.c_int, [_P, C.POINTER(GlfParams), C.POINTER(PlanInfo)]),
    "sta_glf_consensus": (C.c_int, [C.POINTER(GlfCol), C.c_char, C.c_char_p]),
    "sta_calmd_plan": (C.c_int, [_P, C.POINTER(CalmdParams), C.POINTER(PlanInfo)]),
    "sta_fetch_calmd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P]),
    "sta_fetch_calmd_mapq_cap": (C.c_int, [_P, _P]),
    "sta_main_calmd": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "sta_main_glf": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "sta_consensus_run": (C.c_int, [_P, C.POINTER(ConsParams), C.POINTER(ConsInfo)]),
    "sta_fetch_consensus": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "sta_main_consensus": (C.c_int, [C.c_int, C.POINTER(C.c_char_p)]),
    "sta_cons_entries_run": (C.c_int, [_P, C.POINTER(ConsInfo)]),
    "sta_fetch_cons_entries": (C.c_int, [_P, _P, _P, _P, _P, _P, _P]),
    "sta_io_scan": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "sta_io_scan_region": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "sta_format_aux_float": (C.c_int, [C.c_double, C.c_char_p, C.c_int]),
    "sta_io_write_sam": (C.c_int, [C.c_char_p, C.c_char_p]),
    "sta_io_write_bam": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int]),
    "sta_io_fasta_scan": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "sta_cov_hist_begin": (C.c_int, [_P, C.c_int32]),
    "sta_cov_hist_fetch": (C.c_int, [_P, _P, C.c_int32]),
}
EXPORTED_SYMBOLS = sorted(_PROTOS)
for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)   # AttributeError here = the library does not export what the header declares
    _f.restype = _res
    _f.argtypes = _args


def baq_stream_bytes_per_base():
    """forward-row bytes the band-7 BAQ kernel pair streams through HBM per query base (written once, read once)"""
    f = lib.sta_baq_stream_bytes_per_base
    f.restype = C.c_double
    return float(f())


def baq7s_stream_bytes_per_base():
    """the same for the class-S kernel (one row of three stored): bytes written per query base; the launch reads them back once"""
    f = lib.sta_baq7s_stream_bytes_per_base
    f.restype = C.c_double
    return float(f())


def device_count():
    return int(lib.sta_device_count())


def version():
    return lib.sta_version().decode()


class EngineError(RuntimeError):
    pass


def _argv(args):
    arr = (C.c_char_p * (len(args) + 1))()
    for i, a in enumerate(args):
        arr[i] = a.encode()
    return arr


def main_mpileup(args):
    """Run the `mpileup` driver in-process; args excludes the sub-command name."""
    a = ["mpileup"] + list(args)
    return lib.sta_main_mpileup(len(a), _argv(a))


def main_capture(sub, args):
    """(exit status, text bytes) of `sub args...` ("mpileup" / "depth") with the driver's output captured in memory."""
    a = [sub] + list(args)
    buf = _P()
    n = C.c_uint64(0)
    rc = lib.sta_main_capture(len(a), _argv(a), C.byref(buf), C.byref(n))
    try:
        data = C.string_at(buf, n.value) if buf and n.value else b""
    finally:
        if buf:
            lib.sta_capture_free(buf)
    return rc, data


def main_capture_device(sub, args):
    """(exit status, host text bytes, device byte count): the command's window text stays in device memory until
    capture_device_take() copies it to a device pointer; what the driver wrote itself (depth -H's header) is the host text, which
    precedes the windows in the output."""
    a = [sub] + list(args)
    buf = _P()
    n_host, n_dev = C.c_uint64(0), C.c_uint64(0)
    rc = lib.sta_main_capture_device(len(a), _argv(a), C.byref(n_dev), C.byref(buf), C.byref(n_host))
    try:
        head = C.string_at(buf, n_host.value) if buf and n_host.value else b""
    finally:
        if buf:
            lib.sta_capture_free(buf)
    return rc, head, n_dev.value


def capture_device_take(dev_ptr, capacity):
    rc = lib.sta_capture_device_take(dev_ptr, capacity)
    if rc != 0:
        raise RuntimeError("sta_capture_device_take failed: %d" % rc)


def bgzf_inflate(streams, sizes, device=0):
    """Raw deflate streams (bytes objects) -> (list of inflated bytes or None where the device gave up, status list, kernel ms):
    every stream is one block of sta_bgzf_inflate (csrc/kernels_inflate.hip, one wave per block)."""
    import numpy as np

    class Blk(C.Structure):
        _fields_ = [("comp_off", C.c_uint64), ("clen", C.c_uint32), ("isize", C.c_uint32), ("out_off", C.c_uint64)]
    n = len(streams)
    comp = b"".join(streams)
    blocks = (Blk * max(n, 1))()
    co = oo = 0
    for i, (st, sz) in enumerate(zip(streams, sizes)):
        blocks[i].comp_off = co; blocks[i].clen = len(st); blocks[i].isize = sz; blocks[i].out_off = oo
        co += len(st); oo += sz
    cbuf = np.frombuffer(comp if comp else b"\0", dtype=np.uint8)
    out = np.zeros(max(oo, 1), dtype=np.uint8)
    status = np.zeros(max(n, 1), dtype=np.uint32)
    ms = C.c_double(0.0)
    rc = lib.sta_bgzf_inflate(device, cbuf.ctypes.data, len(comp), C.addressof(blocks), n, out.ctypes.data, oo, status.ctypes.data, C.byref(ms))
    if rc != 0:
        raise RuntimeError("sta_bgzf_inflate failed: %d" % rc)
    res, o = [], 0
    for i, sz in enumerate(sizes):
        res.append(out[o:o + sz].tobytes() if status[i] == 0 else None)
        o += sz
    return res, [int(x) for x in status[:n]], ms.value


def io_scan_region(path, region, threads=0, use_index=True):
    """(records, checksum, used_index) of one region of a BAM, read the way a `-r` run reads it; needs no device."""
    n = C.c_uint64(0); h = C.c_uint64(0); u = C.c_int(0)
    rc = lib.sta_io_scan_region(path.encode(), region.encode(), threads, 1 if use_index else 0, C.byref(n), C.byref(h), C.byref(u))
    if rc != 0:
        raise RuntimeError("sta_io_scan_region(%s, %s) failed: %d" % (path, region, rc))
    return n.value, h.value, bool(u.value)


def io_write_sam(path, out_path):
    """every record of a SAM/BAM file written back as SAM text by the drivers' reader + record formatter; needs no device."""
    rc = lib.sta_io_write_sam(os.fsencode(path), os.fsencode(out_path))
    if rc != 0:
        raise RuntimeError("sta_io_write_sam(%s) failed: %d" % (path, rc))


def io_write_bam(path, out_path, level=6):
    """the same as BAM: level 0 = stored BGZF blocks (calmd -u), otherwise compressed (-b); needs no device."""
    rc = lib.sta_io_write_bam(os.fsencode(path), os.fsencode(out_path), int(level))
    if rc != 0:
        raise RuntimeError("sta_io_write_bam(%s) failed: %d" % (path, rc))


def io_fasta_scan(path, order=0):
    """(contigs, bases, checksum, used_index) of a reference FASTA read through the drivers' loader; needs no device."""
    n, b, h, z = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_int(0)
    rc = lib.sta_io_fasta_scan(os.fsencode(path), int(order), C.byref(n), C.byref(b), C.byref(h), C.byref(z))
    if rc != 0:
        raise RuntimeError("sta_io_fasta_scan(%s) failed: %d" % (path, rc))
    return n.value, b.value, h.value, bool(z.value)


def main_depth(args):
    a = ["depth"] + list(args)
    return lib.sta_main_depth(len(a), _argv(a))


def io_scan(path, threads=0, stage=False):
    """(records, checksum) of a SAM/BAM file decoded by the drivers' reader; needs no device.
    stage: False/0 = records only, True/1 = + window pump and stager (record lane), 2 = chunk lane."""
    n, h = C.c_uint64(0), C.c_uint64(0)
    if isinstance(path, (list, tuple)):
        path = "\n".join(path)          # several inputs = the drivers' multi-file windows
    rc = lib.sta_io_scan(os.fsencode(path), int(threads), int(stage), C.byref(n), C.byref(h))
    if rc != 0:
        raise RuntimeError("sta_io_scan(%s) failed: %d" % (path, rc))
    return n.value, h.value


class Engine:
    """Owns one sta_engine.  `stream` is a raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""

    def __init__(self, device=0, stream=None):
        self._h = _P()
        rc = lib.sta_engine_create(C.byref(self._h), int(device), _P(stream or 0))
        if rc == STA_ERR_NO_DEVICE:
            raise EngineError("no usable HIP device %d (the engine has no CPU fallback)" % device)
        if rc != STA_OK:
            raise EngineError("sta_engine_create failed: %d" % rc)
        self._keep = []

    def close(self):
        if self._h:
            lib.sta_engine_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != STA_OK:
            raise EngineError("%s failed (%d): %s" % (what, rc, lib.sta_last_error(self._h).decode()))

    def set_reference(self, tid, ptr, length, mem=STA_MEM_HOST):
        self._chk(lib.sta_set_reference(self._h, tid, _P(ptr), length, mem), "sta_set_reference")

    def clear_references(self):
        lib.sta_clear_references(self._h)

    def stage_window(self, window):
        self._keep = [window]
        self._chk(lib.sta_stage_window(self._h, C.byref(window)), "sta_stage_window")

    def mpileup_plan(self, params):
        info = PlanInfo()
        self._chk(lib.sta_mpileup_plan(self._h, C.byref(params), C.byref(info)), "sta_mpileup_plan")
        return info

    def mpileup_emit(self, dev_ptr=None, capacity=0):
        self._chk(lib.sta_mpileup_emit(self._h, _P(dev_ptr or 0), capacity), "sta_mpileup_emit")

    def mpileup_run(self, params, dev_ptr=None, capacity=0):
        """plan + emit in one call (single-pass kernel); text in dev_ptr, or in the engine's buffer when None"""
        info = PlanInfo()
        self._chk(lib.sta_mpileup_run(self._h, C.byref(params), _P(dev_ptr or 0), capacity, C.byref(info)), "sta_mpileup_run")
        return info

    def depth_plan(self, params):
        info = PlanInfo()
        self._chk(lib.sta_depth_plan(self._h, C.byref(params), C.byref(info)), "sta_depth_plan")
        return info

    def depth_run(self, params, dev_ptr=None, capacity=0):
        info = PlanInfo()
        self._chk(lib.sta_depth_run(self._h, C.byref(params), _P(dev_ptr or 0), capacity, C.byref(info)), "sta_depth_run")
        return info

    def depth_emit(self, dev_ptr=None, capacity=0):
        self._chk(lib.sta_depth_emit(self._h, _P(dev_ptr or 0), capacity), "sta_depth_emit")

    def glf_plan(self, min_baseQ=13, max_depth=8000, theta=0.83):
        """bcf_call_glfgen over every column of the staged window; results via fetch_output (GlfCol[col][file])."""
        info, p = PlanInfo(), GlfParams(min_baseQ, max_depth, theta)
        self._chk(lib.sta_glf_plan(self._h, C.byref(p), C.byref(info)), "sta_glf_plan")
        return info

    def consensus_run(self, params):
        """`samtools consensus` columns of file 0 of the staged window (sta_consensus_run)"""
        info = ConsInfo()
        self._chk(lib.sta_consensus_run(self._h, C.byref(params), C.byref(info)), "sta_consensus_run")
        return info

    def fetch_consensus(self, n_positions, info, want_text=False):
        """-> (ins[n_positions], cols[n_cols], col_off, seq_chars, qual_chars); the last three are None without want_text"""
        ins = (C.c_int32 * max(1, n_positions))()
        cols = (ConsCol * max(1, info.n_cols))()
        off = (C.c_uint64 * (info.n_cols + 1))() if want_text else None
        sq = C.create_string_buffer(max(1, info.n_entries)) if want_text else None
        ql = C.create_string_buffer(max(1, info.n_entries)) if want_text else None
        self._chk(lib.sta_fetch_consensus(self._h, C.cast(ins, _P), C.cast(cols, _P), C.cast(off, _P) if want_text else None,
                                          C.cast(sq, _P) if want_text else None, C.cast(ql, _P) if want_text else None), "sta_fetch_consensus")
        return ins, cols, off, sq, ql

    def calmd_plan(self, flag=0, max_nm=0, capQ=0):
        """calmd's MD / NM / BAQ-tag arithmetic on file 0 of the staged window; info.out_bytes = MD text bytes."""
        info, p = PlanInfo(), CalmdParams(flag, max_nm, capQ)
        self._chk(lib.sta_calmd_plan(self._h, C.byref(p), C.byref(info)), "sta_calmd_plan")
        return info

    def depth_counts_ptr(self):
        return lib.sta_depth_counts_dev(self._h)

    def fetch_output(self, nbytes):
        buf = C.create_string_buffer(int(nbytes) if nbytes else 1)
        self._chk(lib.sta_fetch_output(self._h, C.cast(buf, _P), int(nbytes)), "sta_fetch_output")
        return buf.raw[:int(nbytes)]

    def fetch_col_offsets(self, n):
        """first n (<= columns + 1) exclusive column offsets of the planned window, in bytes of the window's text"""
        arr = (C.c_uint64 * int(n))()
        self._chk(lib.sta_fetch_col_offsets(self._h, C.cast(arr, _P), int(n)), "sta_fetch_col_offsets")
        return list(arr)

    def sync(self):
        self._chk(lib.sta_sync(self._h), "sta_sync")

    def profile(self, on=True):
        lib.sta_profile_enable(self._h, 1 if on else 0)

    def profile_only(self, name=None):
        lib.sta_profile_only(self._h, name.encode() if name else None)

    def profile_reset(self):
        lib.sta_profile_reset(self._h)

    def profile_get(self):
        arr = (KernelTime * 64)()
        n = lib.sta_profile_get(self._h, arr, 64)
        return {arr[i].name.decode(): (int(arr[i].launches), float(arr[i].total_ms)) for i in range(min(n, 64))}
