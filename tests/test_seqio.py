"""FASTA / PHYLIP readers (libpll_amd/csrc/host/seqio.c) against the reference's
fasta.c / phylip.c as built into oracle/_ref/libpll_ref.so: the same files through
both, every record, counter, return value, pll_errno and pll_errmsg compared.
Covers what test/src/00110_NPDN_fasta.c and 00120_NPAN_fasta.c exercise (their
data files are not part of the reference snapshot) plus malformed inputs.
CPU-only: no device is touched."""
import ctypes as C
import os

import numpy as np
import pytest


class Fasta(C.Structure):
    _fields_ = [("fp", C.c_void_p), ("line", C.c_char * 2048), ("chrstatus", C.c_void_p),
                ("no", C.c_long), ("filesize", C.c_long), ("lineno", C.c_long),
                ("stripped_count", C.c_long), ("stripped", C.c_long * 256)]


class Phylip(C.Structure):
    _fields_ = [("fp", C.c_void_p), ("line", C.c_void_p), ("line_size", C.c_size_t),
                ("line_maxsize", C.c_size_t), ("buffer", C.c_char * 2048), ("chrstatus", C.c_void_p),
                ("no", C.c_long), ("filesize", C.c_long), ("lineno", C.c_long),
                ("stripped_count", C.c_long), ("stripped", C.c_long * 256)]


class Msa(C.Structure):
    _fields_ = [("count", C.c_int), ("length", C.c_int), ("sequence", C.POINTER(C.c_char_p)),
                ("label", C.POINTER(C.c_char_p))]


libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]


def bind(lib):
    L = lib.lib
    L.pll_fasta_open.restype = C.POINTER(Fasta)
    L.pll_fasta_open.argtypes = [C.c_char_p, C.c_void_p]
    L.pll_fasta_getnext.argtypes = [C.POINTER(Fasta), C.POINTER(C.c_void_p), C.POINTER(C.c_long),
                                    C.POINTER(C.c_void_p), C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.pll_fasta_close.argtypes = [C.POINTER(Fasta)]
    L.pll_fasta_rewind.argtypes = [C.POINTER(Fasta)]
    L.pll_fasta_getfilesize.restype = C.c_long
    L.pll_fasta_getfilesize.argtypes = [C.POINTER(Fasta)]
    L.pll_fasta_getfilepos.restype = C.c_long
    L.pll_fasta_getfilepos.argtypes = [C.POINTER(Fasta)]
    L.pll_phylip_open.restype = C.POINTER(Phylip)
    L.pll_phylip_open.argtypes = [C.c_char_p, C.c_void_p]
    L.pll_phylip_close.argtypes = [C.POINTER(Phylip)]
    L.pll_phylip_rewind.argtypes = [C.POINTER(Phylip)]
    for f in (L.pll_phylip_parse_interleaved, L.pll_phylip_parse_sequential):
        f.restype = C.POINTER(Msa)
        f.argtypes = [C.POINTER(Phylip)]
    L.pll_msa_destroy.argtypes = [C.POINTER(Msa)]
    return L


def err(lib):
    return (C.c_int.in_dll(lib.lib, "pll_errno").value, lib.errmsg())


def set_errno(lib, v):
    C.c_int.in_dll(lib.lib, "pll_errno").value = v


def read_fasta(lib, path, rewind_after=None):
    """Everything observable from reading `path` record by record."""
    L = bind(lib)
    table = C.addressof((C.c_uint * 256).in_dll(lib.lib, "pll_map_fasta"))
    set_errno(lib, 0)
    fd = L.pll_fasta_open(path.encode(), table)
    if not fd:
        return ("open failed",) + err(lib)
    out = [("size", L.pll_fasta_getfilesize(fd))]
    passes = 2 if rewind_after is not None else 1
    for _ in range(passes):
        n = 0
        while True:
            head, seq = C.c_void_p(), C.c_void_p()
            hl, sl, no = C.c_long(), C.c_long(), C.c_long(-7)
            ok = L.pll_fasta_getnext(fd, C.byref(head), C.byref(hl), C.byref(seq), C.byref(sl), C.byref(no))
            if not ok:
                out.append(("stop",) + err(lib) + (fd.contents.lineno,))
                break
            out.append((C.string_at(head.value), hl.value, C.string_at(seq.value), sl.value, no.value,
                        L.pll_fasta_getfilepos(fd), fd.contents.lineno))
            libc.free(head)
            libc.free(seq)
            n += 1
            if rewind_after is not None and n == rewind_after and _ == 0:
                out.append(("rewind", L.pll_fasta_rewind(fd), fd.contents.lineno))
                break
        out.append(("stripped", fd.contents.stripped_count, tuple(fd.contents.stripped)))
    L.pll_fasta_close(fd)
    return out


def read_phylip(lib, path, interleaved, twice=False):
    L = bind(lib)
    table = C.addressof((C.c_uint * 256).in_dll(lib.lib, "pll_map_phylip"))
    set_errno(lib, 0)
    fd = L.pll_phylip_open(path.encode(), table)
    if not fd:
        return ("open failed",) + err(lib)
    out = [("size", fd.contents.filesize)]
    for rep in range(2 if twice else 1):
        if rep:
            out.append(("rewind", L.pll_phylip_rewind(fd)))
        set_errno(lib, 0)
        msa = (L.pll_phylip_parse_interleaved if interleaved else L.pll_phylip_parse_sequential)(fd)
        if not msa:
            e = err(lib)
            # a refused header leaves pll_errno alone and pll_errmsg stale: compare the code only
            out.append(("failed", e[0], e[1] if e[0] else ""))
            break
        m = msa.contents
        out.append((m.count, m.length, [m.label[i] for i in range(m.count)],
                    [m.sequence[i] for i in range(m.count)]))
        out.append(("stripped", fd.contents.stripped_count, tuple(fd.contents.stripped), fd.contents.lineno))
        L.pll_msa_destroy(msa)
    L.pll_phylip_close(fd)
    return out


def rng_seq(rng, n, alphabet):
    return "".join(alphabet[i] for i in rng.integers(0, len(alphabet), n))


def fasta_corpus(tmp):
    rng = np.random.default_rng(4)
    nt, aa = "ACGTacgtNRYKM-", "ARNDCQEGHILKMFPSTWYV-X*"
    files = {}

    def put(name, text, binary=False):
        p = os.path.join(tmp, name)
        with open(p, "wb") as f:
            f.write(text if binary else text.encode())
        files[name] = p

    put("plain.fas", "".join(">seq%d some description\n%s\n" % (i, rng_seq(rng, 70, nt)) for i in range(5)))
    put("wrapped.fas", "".join(">s%d\n%s" % (i, "".join(rng_seq(rng, 60, aa) + "\n" for _ in range(7)))
                               for i in range(4)))
    put("crlf.fas", ">a desc\r\nACGT\r\nAC-T\r\n>b\r\nGGGG\r\n")
    put("blanks_digits.fas", ">x\nAC GT 12 AC\n\n\tGT\n>y\n  \nAAAA\n")
    put("no_trailing_newline.fas", ">a\nACGT\n>b\nACGA")
    put("empty_sequence.fas", ">a\n>b\nACGT\n>c\n")
    put("long_line.fas", ">a\n" + rng_seq(rng, 9000, nt) + "\n>b\n" + rng_seq(rng, 2047, nt) + "\n")
    put("long_header.fas", ">" + "h" * 5000 + "\nACGT\n")
    put("header_len_2046.fas", ">" + "h" * 2046 + "\nACGT\n")
    put("bad_header.fas", "ACGT\n>a\nACGT\n")
    put("second_header_bad.fas", ">a\nACGT\n")          # control
    put("illegal_char.fas", ">a\nACGT\n>b\nAC#T\nAAAA\n")
    put("unprintable.fas", b">a\nAC\x01GT\n", binary=True)
    put("empty.fas", "")
    put("only_newline.fas", "\n")
    put("many.fas", "".join(">t%04d\n%s\n" % (i, rng_seq(rng, int(rng.integers(1, 300)), nt))
                            for i in range(300)))
    put("lowercase_and_symbols.fas", ">q\nacgu.~?!ACGU\n")
    return files


def phylip_corpus(tmp):
    rng = np.random.default_rng(9)
    nt = "ACGTNRY-"
    files = {}

    def put(name, text):
        p = os.path.join(tmp, name)
        with open(p, "w") as f:
            f.write(text)
        files[name] = p

    seqs = [rng_seq(rng, 120, nt) for _ in range(6)]
    put("seq_oneline.phy", "6 120\n" + "".join("taxon%d %s\n" % (i, s) for i, s in enumerate(seqs)))
    put("seq_wrapped.phy", " 6   120 \n" + "".join(
        "t%d\t%s\n%s\n%s\n" % (i, s[:50], s[50:100], s[100:]) for i, s in enumerate(seqs)))
    put("seq_spaces_in_data.phy", "6 120\n" + "".join(
        "name_%d   %s\n" % (i, " ".join(s[j:j + 10] for j in range(0, 120, 10))) for i, s in enumerate(seqs)))
    inter = "6 120\n" + "".join("tx%d  %s\n" % (i, s[:40]) for i, s in enumerate(seqs)) + "\n" + \
            "".join("%s\n" % s[40:80] for s in seqs) + "\n\n" + "".join("   %s\n" % s[80:] for s in seqs)
    put("interleaved.phy", inter)
    put("interleaved_no_final_newline.phy", inter.rstrip("\n"))
    put("interleaved_option_letter.phy", inter.replace("6 120", "6 120 I", 1))
    put("header_trailing_text.phy", "6 120 extra\n" + "".join("t%d %s\n" % (i, s) for i, s in enumerate(seqs)))
    put("bad_count.phy", "x 120\nA ACGT\n")
    put("zero_length.phy", "2 0\nA ACGT\nB ACGT\n")
    put("missing_length.phy", "2\nA ACGT\nB ACGT\n")
    put("too_few.phy", "3 8\nA ACGTACGT\nB ACGTACGT\n")
    put("too_many.phy", "2 8\nA ACGTACGT\nB ACGTACGT\nC ACGTACGT\n")
    put("long_seq.phy", "2 8\nA ACGTACGTAA\nB ACGTACGT\n")
    put("short_seq.phy", "2 8\nA ACGTACGT\nB ACGT\n")
    put("illegal_char.phy", "2 8\nA ACGT#CGT\nB ACGTACGT\n")
    put("nonaligned_block.phy", "2 8\nA ACGT\nB ACG\n\nACGT\nACGTA\n")
    put("partial_last_block.phy", "2 8\nA ACGT\nB ACGT\n\nACGT\n")
    put("long_line.phy", "2 5000\nfirst " + rng_seq(rng, 5000, nt) + "\nsecond " + rng_seq(rng, 5000, nt) + "\n")
    put("label_tab.phy", "2 4\nalpha\tACGT\nbeta\tAC-T\n")
    put("label_only_line.phy", "2 4\nalpha\nACGT\nbeta\nACGT\n")
    put("blank_lines.phy", "2 4\n\n\nA ACGT\n\n   \nB ACGT\n\n")
    put("empty.phy", "")
    put("header_only.phy", "3 10\n")
    return files


@pytest.fixture(scope="module")
def libs(amd, ref):
    return amd, ref


def test_fasta_matches_reference(libs, tmp_path):
    amd, ref = libs
    files = fasta_corpus(str(tmp_path))
    assert len(files) >= 15
    for name, path in sorted(files.items()):
        got, want = read_fasta(amd, path), read_fasta(ref, path)
        assert got == want, name
    for name in ("plain.fas", "many.fas"):
        assert read_fasta(amd, files[name], rewind_after=2) == read_fasta(ref, files[name], rewind_after=2)
    missing = os.path.join(str(tmp_path), "unexistent-file")
    assert read_fasta(amd, missing) == read_fasta(ref, missing)
    assert read_fasta(amd, missing)[1] == 100    # PLL_ERROR_FILE_OPEN (test 00110's first case)


@pytest.mark.parametrize("interleaved", [False, True], ids=["sequential", "interleaved"])
def test_phylip_matches_reference(libs, tmp_path, interleaved):
    amd, ref = libs
    files = phylip_corpus(str(tmp_path))
    ok = 0
    for name, path in sorted(files.items()):
        got, want = read_phylip(amd, path, interleaved), read_phylip(ref, path, interleaved)
        assert got == want, name
        ok += isinstance(got, list) and len(got) > 1 and got[1][0] != "failed"
    assert ok >= 5, "corpus no longer has alignments this layout can read"
    # rewind + second parse: the reference dereferences a released line buffer here
    # (phylip.c:131-136 frees it at end of file, :117-123 then copies into it), so
    # only the product is exercised: the second pass must repeat the first
    p = files["interleaved.phy" if interleaved else "seq_wrapped.phy"]
    both = read_phylip(amd, p, interleaved, twice=True)
    assert both[3] == ("rewind", 1) and both[4] == both[1] and both[5] == both[2]


def test_fasta_into_partition_roundtrip(amd, tmp_path):
    """The reader's output is what pll_set_tip_states takes (the examples/ flow):
    header -> tip name, sequence -> tip states; here just the text contract."""
    path = os.path.join(str(tmp_path), "x.fas")
    rows = {"tipA": "ACGTAC-T", "tipB": "ACGTRYAC", "tipC": "NNGTACGT"}
    with open(path, "w") as f:
        for k, v in rows.items():
            f.write(">%s\n%s\n%s\n" % (k, v[:3], v[3:]))
    recs = [r for r in read_fasta(amd, path) if isinstance(r[0], bytes)]
    assert [(r[0].decode(), r[2].decode()) for r in recs] == list(rows.items())
