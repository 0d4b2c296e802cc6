"""Bench-size runs through the C-ABI (ctypes, device-resident inputs) checked with size-independent properties
instead of a byte-for-byte oracle diff (the oracle needs minutes at this size):
  * depth: binary per-column counts == numpy difference-array recount of the same reads (exact);
  * mpileup -B: every column of the window appears once and in order, the count column sums to the number of
    (read, column) pairs whose base quality passes -Q13 (numpy recount), and the text length satisfies the
    per-line identity bytes = fixed + 2*count + head/tail marks;
  * mpileup with BAQ: idempotence of the plan (two plans of the same staged window give identical text) and
    monotonicity (BAQ only lowers qualities, so every column's count <= the -B count).
Needs a GPU: -m gpu."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_COLS = 1 << 21          # 2 Mi columns, 30x, 150 bp: 419 430 reads, 62.9 M piled bases


def _stage(sa, torch, rd, ref, n_cols, dev):
    keep = {}

    def up(name):
        arr = rd[name]
        if arr.dtype == np.uint32: arr = arr.view(np.int32)
        elif arr.dtype == np.uint16: arr = arr.view(np.int16)
        elif arr.dtype == np.uint64: arr = arr.view(np.int64)
        t = torch.from_numpy(np.ascontiguousarray(arr).copy()).to(dev)
        keep[name] = t
        return t.data_ptr()

    reads = sa.Reads()
    reads.n_reads = rd["n"]
    for f in ("pos", "flag", "mapq", "aux", "l_qseq", "cig_off", "base_off8", "mtid", "mpos", "isize", "name_off", "cigar", "seq", "qual", "names"):
        setattr(reads, f, up(f))
    reads.bq = None
    reads.n_cigar_total = len(rd["cigar"]); reads.n_bases_total = len(rd["qual"]); reads.n_name_bytes = len(rd["names"])
    files = (sa.Reads * 1)(reads)
    w = sa.Window()
    w.tid = 0; w.origin = 0; w.col_beg = 0; w.col_end = n_cols
    w.tname = b"chrS"; w.tlen = n_cols; w.n_files = 1; w.files = files; w.mem = 1
    w.has_bed = 0; w.has_reg = 0
    keep["files"] = files
    return w, keep


@pytest.fixture(scope="module")
def big():
    import torch
    import samtools_amd as sa
    from synth import synth_ref, synth_reads
    dev = torch.device("cuda", 0)
    ref = synth_ref(N_COLS, seed=11)
    rd = synth_reads(ref, depth=30, read_len=150, seed=12, indel_rate=0.0)      # pure 150M reads: closed-form recounts
    eng = sa.Engine(0, torch.cuda.current_stream().cuda_stream)
    ref_t = torch.from_numpy(ref.copy()).to(dev)
    eng.set_reference(0, ref_t.data_ptr(), N_COLS, 1)
    w, keep = _stage(sa, torch, rd, ref, N_COLS, dev)
    return dict(torch=torch, sa=sa, eng=eng, w=w, keep=keep, rd=rd, ref_t=ref_t, dev=dev)


def _numpy_depth(rd, n_cols, min_q=None):
    pos = rd["_abs_pos"].astype(np.int64)
    L = rd["L"]
    if min_q is None:
        d = np.zeros(n_cols + 1, dtype=np.int64)
        np.add.at(d, pos, 1); np.add.at(d, pos + L, -1)
        return np.cumsum(d)[:n_cols]
    ok = rd["_quals"] >= min_q
    cols = (pos[:, None] + np.arange(L)[None, :])[ok]
    return np.bincount(cols, minlength=n_cols)[:n_cols]


def test_depth_counts_at_bench_size_equal_numpy(big):
    sa, eng, torch = big["sa"], big["eng"], big["torch"]
    par = sa.DepthParams.defaults(); par.all_pos = 1
    eng.stage_window(big["w"])
    info = eng.depth_plan(par)
    ptr = eng.depth_counts_ptr()
    n = N_COLS + 1
    # counts row 0 lives in engine-owned device memory: device-to-device copy into a torch tensor, then to the host
    t = torch.empty(n, dtype=torch.int32, device=big["dev"])
    C.cdll.LoadLibrary("libamdhip64.so").hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * 4), 3)
    got = t.cpu().numpy()[:N_COLS].astype(np.int64)
    want = _numpy_depth(big["rd"], N_COLS)
    assert int(info.piled_bases) == int(big["rd"]["n"]) * 150
    assert np.array_equal(got, want)
    assert int(info.n_lines) == N_COLS            # -a: one row per position


def _parse_counts(text):
    """count column (4th field) and line lengths of mpileup text, vectorised."""
    b = np.frombuffer(text, dtype=np.uint8)
    nl = np.flatnonzero(b == 10)
    starts = np.concatenate(([0], nl[:-1] + 1))
    tabs = np.flatnonzero(b == 9)
    # every line has exactly 5 tabs (no extra columns): the 3rd tab of line k is tabs[5k+2], the 4th tabs[5k+3]
    assert len(tabs) == 5 * len(nl)
    t3, t4 = tabs[2::5], tabs[3::5]
    width = t4 - t3 - 1
    counts = np.zeros(len(nl), dtype=np.int64)
    for k in range(int(width.max())):
        sel = width > k
        digit = b[t3[sel] + 1 + k].astype(np.int64) - 48
        counts[sel] = counts[sel] * 10 + digit
    # 2nd field = position
    t1, t2 = tabs[0::5], tabs[1::5]
    pw = t2 - t1 - 1
    posv = np.zeros(len(nl), dtype=np.int64)
    for k in range(int(pw.max())):
        sel = pw > k
        posv[sel] = posv[sel] * 10 + (b[t1[sel] + 1 + k].astype(np.int64) - 48)
    return posv, counts, nl - starts + 1, (t4, tabs[4::5], nl)


def _mpileup_text(big, realn):
    sa, eng = big["sa"], big["eng"]
    par = sa.MplpParams.defaults(); par.has_fai = 1
    if not realn:
        par.flag &= ~sa.MPLP.REALN
    eng.stage_window(big["w"])
    info = eng.mpileup_plan(par)
    eng.mpileup_emit()
    return info, eng.fetch_output(int(info.out_bytes))


def test_mpileup_B_properties_at_bench_size(big):
    info, text = _mpileup_text(big, realn=False)
    posv, counts, linelen, (t4, t5, nl) = _parse_counts(text)
    want = _numpy_depth(big["rd"], N_COLS, min_q=13)
    cov = _numpy_depth(big["rd"], N_COLS) > 0
    assert len(posv) == int(cov.sum()) == int(info.n_lines)          # every covered column once
    assert np.array_equal(posv, np.flatnonzero(cov) + 1)              # ... in order, 1-based
    assert np.array_equal(counts, want[cov])                          # -Q13 filter, exact recount
    # quality string length == count (or 1 for '*'); base string >= count
    qlen = nl - t5 - 1
    assert np.array_equal(qlen, np.maximum(counts, 1))
    slen = t5 - t4 - 1
    assert np.all(slen >= np.maximum(counts, 1))
    assert int(linelen.sum()) == len(text) == int(info.out_bytes)


def test_mpileup_baq_idempotent_and_monotone_at_bench_size(big):
    info_b, text_b = _mpileup_text(big, realn=False)
    info1, text1 = _mpileup_text(big, realn=True)
    info2, text2 = _mpileup_text(big, realn=True)
    assert text1 == text2                                              # staging is not modified by a plan
    _, c_b, _, _ = _parse_counts(text_b)
    p1, c_1, _, _ = _parse_counts(text1)
    assert len(c_b) == len(c_1)
    assert np.all(c_1 <= c_b)                                          # BAQ only lowers base qualities
    assert int(c_1.sum()) < int(c_b.sum())


def test_col_offsets_are_window_offsets_of_the_emitted_rows(big):
    """sta_fetch_col_offsets after a text plan: offsets inside the WHOLE window's text (include/samtools_amd.h), also on the tile
    path, whose device-side offsets restart every 1 024 columns: the offsets must be the row starts of the text."""
    info, text = _mpileup_text(big, realn=False)
    eng = big["eng"]
    offs = np.array(eng.fetch_col_offsets(N_COLS + 1), dtype=np.int64)
    assert offs[0] == 0 and int(offs[-1]) == len(text) == int(info.out_bytes)
    assert np.all(np.diff(offs) >= 0)
    nl = np.flatnonzero(np.frombuffer(text, dtype=np.uint8) == 10)
    starts = np.concatenate(([0], nl[:-1] + 1))
    cov = _numpy_depth(big["rd"], N_COLS) > 0                          # without -a a row exists only for covered columns
    assert len(starts) == int(cov.sum())
    assert np.array_equal(offs[:-1][cov], starts)
    assert np.array_equal(np.diff(offs)[~cov], np.zeros(int((~cov).sum()), dtype=np.int64))
