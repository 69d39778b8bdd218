"""Static ABI check of the Julia shim.  There is no `julia` binary in the build image, so
bridge.jl_amd/julia/BridgeHIP.jl has never been parsed; what CAN be checked without Julia is that every
`ccall((:name, lib), Ret, (ArgTypes...), values...)` in it agrees with the prototype of `name` in include/bridgehip.h:
the symbol exists (in the header and in the built library), the return type, the number of argument types, the number of
values handed over, and each argument's type under the map

    Cint<->int  Clong<->long  Csize_t<->size_t  Cdouble<->double  Cfloat<->float  UInt32<->uint32_t  UInt64<->uint64_t
    Int64<->int64_t  Cstring<->const char*  Ptr{T} / Ref{T} <-> T* (same pointer depth; Cvoid / UInt8 match any pointee)

plus the converse: the header's entry points the shim does NOT bind, as an explicit allow-list (a new entry point must
either be bound or be added here on purpose).  The shim extends the dispatch surface of src/euler.jl:11-63,247-268 and
src/types.jl:23 (SDESolver / ContinuousTimeProcess)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "bridge.jl_amd", "julia", "BridgeHIP.jl")
HEADER = os.path.join(ROOT, "include", "bridgehip.h")

SCALARS = {"Cint": "int", "Clong": "long", "Csize_t": "size_t", "Cdouble": "double", "Cfloat": "float", "UInt32": "uint32_t",
           "UInt64": "uint64_t", "Int64": "int64_t", "UInt8": "uint8_t", "Cvoid": "void", "Cchar": "char"}
ANY_POINTEE = {"void", "uint8_t"}          # Ptr{Cvoid} / Ptr{UInt8}: opaque handles, device pointers, byte buffers

# header entry points the Julia shim leaves unbound -- each for a reason
NOT_BOUND = {
    # plumbing the shim does not need (Julia has its own): versions, counts, memset, host->device copies of raw bytes
    "bhip_version", "bhip_device_count", "bhip_ctx_sync", "bhip_memcpy_h2d", "bhip_memset", "bhip_upload_aos",
    # proposal construction variants the shim reaches through bhip_proposal_set_aux_callback (a Julia @cfunction over Bridge.B / Bridge.β / Bridge.a)
    "bhip_proposal_set_aux", "bhip_proposal_set_aux_linearappr", "bhip_linearappr", "bhip_linearnoiseappr_path",
    "bhip_proposal_set_aux_linearnoiseappr", "bhip_proposal_guide_get", "bhip_proposal_info", "bhip_model_define_components",
    # the host keeps Bridge.jl's own gpupdate / mcstats (src/guip.jl:221-231, src/mclog.jl)
    "bhip_gpupdate", "bhip_welford_merge",
    # hot path variants the Julia methods do not expose: the fused form goes through sample_solve_parts!, innovations stays Bridge.jl's
    "bhip_sample_solve", "bhip_innovations",
    # reached through its one-launch form over the container's buffers (bhip_girsanov_parts; one buffer delegates to it)
    "bhip_girsanov",
    # chains: read-backs beyond ll / acc / stats, checkpointing
    "bhip_chains_get_paths", "bhip_chains_current_X", "bhip_chains_proposal_X", "bhip_chains_pathstats", "bhip_chains_state_bytes",
    "bhip_chains_save", "bhip_chains_load", "bhip_ctx_piece_of",
    # smoothing ensembles: read-backs and the host-driven adaptation
    "bhip_segchains_placement_info", "bhip_segchains_statistics_info", "bhip_segchains_get_paths", "bhip_segchains_current_X",
    "bhip_segchains_mcstats", "bhip_segchains_set_proposals", "bhip_segchains_pooled_stats", "bhip_segchains_set_pi0",
    "bhip_segchains_chain_guide",
    # communicator forms of the one-process-n-devices launch (the shim is one process per GPU), and the SURVEY aliases
    "bhip_comm_init_all", "bhip_comm_info", "bhip_comm_query", "bhip_comm_allgather", "bhip_comm_init", "bhip_allgather_stats",
    "bhip_comm_allgather_group",
    # RNG specification helpers (host; tests)
    "bhip_philox4x32_10", "bhip_normals_host", "bhip_normals_host_spec",
}


# ------------------------------------------------------------------ the header
def header_prototypes(text=None):
    """name -> (return type, [(base type, pointer depth)])"""
    text = open(HEADER).read() if text is None else text
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    text = re.sub(r"typedef[^;]*;", " ", text)
    protos = {}
    for m in re.finditer(r"((?:const\s+)?\w+\s*\**)\s*\b(bhip_\w+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        params = []
        args = args.strip()
        if args and args != "void":
            for a in args.split(","):
                params.append(c_type(a))
        protos[name] = (c_type(ret + " _")[:2] if "*" in ret else (ret.replace("const", "").strip(), 0), params)
    return protos


def c_type(decl):
    """'const double *const *X_parts' -> ('double', 2);  'const uint32_t ctr[4]' -> ('uint32_t', 1);  'bhip_aux_fn fn' -> ('void', 1)"""
    decl = decl.strip()
    depth = decl.count("*") + decl.count("[")
    decl = re.sub(r"\[[^\]]*\]", "", decl).replace("*", " ")
    words = [w for w in decl.split() if w != "const"]
    base = words[0] if len(words) == 1 else " ".join(words[:-1])        # the last word is the parameter's name
    if base == "bhip_aux_fn":                                          # function-pointer typedef
        return ("void", depth + 1)
    if base.startswith("bhip_"):                                       # opaque handle types
        base = "void"
    return (base, depth)


# ------------------------------------------------------------------ the shim
def split_top(s):
    """split at top-level commas (outside (), {}, [] and string literals)"""
    out, cur, depth, i, q = [], "", 0, 0, None
    while i < len(s):
        ch = s[i]
        if q:
            cur += ch
            if ch == "\\":
                cur += s[i + 1]; i += 1
            elif ch == q:
                q = None
        elif ch == '"':
            q = ch; cur += ch
        elif ch in "({[":
            depth += 1; cur += ch
        elif ch in ")}]":
            depth -= 1; cur += ch
        elif ch == "," and depth == 0:
            out.append(cur.strip()); cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        out.append(cur.strip())
    return out


def strip_comments(src):
    out = []
    for line in src.splitlines():
        q, cut = False, len(line)
        for i, ch in enumerate(line):
            if ch == '"' and (i == 0 or line[i - 1] != "\\"):
                q = not q
            elif ch == "#" and not q:
                cut = i
                break
        out.append(line[:cut])
    return "\n".join(out)


def shim_ccalls(text=None):
    """[(name, ret, [argtypes], nvalues, line)]"""
    src = open(SHIM).read() if text is None else text
    src = re.sub(r'"""(.*?)"""', lambda m: '"' + " " * 0 + '"' + "\n" * m.group(0).count("\n"), src, flags=re.S)   # docstrings mention `ccall` in prose
    src = strip_comments(src)
    calls = []
    for m in re.finditer(r"\bccall\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        parts = split_top(src[m.end():i - 1])
        sym = re.match(r"\(\s*:(\w+)\s*,\s*lib\s*\)$", parts[0])
        assert sym, ("ccall target is not (:name, lib)", parts[0])
        tup = parts[2].strip()
        assert tup.startswith("(") and tup.endswith(")"), parts[2]
        calls.append((sym.group(1), parts[1].strip(), split_top(tup[1:-1]), len(parts) - 3, src.count("\n", 0, m.start()) + 1))
    return calls


def julia_type(t):
    """'Ref{Ptr{Cvoid}}' -> ('void', 2);  'Cstring' -> ('char', 1)"""
    t, depth = t.strip(), 0
    while True:
        m = re.match(r"(?:Ptr|Ref)\{(.*)\}$", t)
        if not m:
            break
        t, depth = m.group(1).strip(), depth + 1
    if t == "Cstring":
        return ("char", depth + 1)
    assert t in SCALARS, f"Julia type {t!r} is not in the map"
    return (SCALARS[t], depth)


def compatible(jl, c):
    (jb, jd), (cb, cd) = jl, c
    if jd != cd:
        # Ptr{Cvoid} may stand for a pointer of any depth >= 1 handed through opaquely (void ** as "some pointer")
        return jd == 1 and jb in ANY_POINTEE and cd >= 1
    if jd == 0:
        return jb == cb
    return jb == cb or jb in ANY_POINTEE or (cb == "void" and jd >= 2)


def check_shim(shim_text=None, header_text=None):
    protos = header_prototypes(header_text)
    problems = []
    bound = set()
    for name, ret, argtypes, nvalues, line in shim_ccalls(shim_text):
        where = f"BridgeHIP.jl:{line} {name}"
        if name not in protos:
            problems.append(f"{where}: no such entry point in include/bridgehip.h")
            continue
        bound.add(name)
        cret, cparams = protos[name]
        if julia_type(ret) != cret:
            problems.append(f"{where}: returns {ret}, the header says {cret}")
        if len(argtypes) != len(cparams):
            problems.append(f"{where}: {len(argtypes)} argument types, the header has {len(cparams)}")
            continue
        if nvalues != len(argtypes):
            problems.append(f"{where}: {nvalues} values for {len(argtypes)} argument types")
        for k, (jt, ct) in enumerate(zip(argtypes, cparams)):
            if not compatible(julia_type(jt), ct):
                problems.append(f"{where}: argument {k + 1} is {jt}, the header says {ct[0]}{'*' * ct[1]}")
    return problems, bound, protos


def test_header_parser_sees_every_export():
    """the prototypes parsed out of the header == the bhip_* symbols the library exports (when it is built)"""
    protos = header_prototypes()
    assert len(protos) >= 90 and protos["bhip_ctx_create"] == (("int", 0), [("int", 0), ("void", 1), ("void", 2)])
    assert protos["bhip_last_error"] == (("char", 1), [("void", 1)])
    assert protos["bhip_sample_solve_parts"][1][4] == ("double", 2) and protos["bhip_philox4x32_10"][1][0] == ("uint32_t", 1)
    so = os.path.join(ROOT, "bridge.jl_amd", "libbridgehip.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    lib = ctypes.CDLL(so)
    for name in protos:
        assert hasattr(lib, name), name


def test_every_ccall_of_the_shim_matches_the_header():
    problems, bound, protos = check_shim()
    assert not problems, "\n".join(problems)
    assert len(bound) >= 45


def test_unbound_entry_points_are_the_allow_list():
    _, bound, protos = check_shim()
    unbound = set(protos) - bound
    assert unbound == NOT_BOUND, (sorted(unbound - NOT_BOUND), sorted(NOT_BOUND - unbound))


def test_the_check_turns_red_on_a_drifted_shim():
    """scratch copies with one deliberate error each: arity, a scalar width, a return type, a missing value, an unknown symbol"""
    src = open(SHIM).read()
    ok = "ccall((:bhip_chains_step, lib), Cint, (Ptr{Cvoid}, Cdouble, Cint, Cint), ch.h, ρ, iterations, skip)"
    assert ok in src
    for bad, what in ((ok.replace("(Ptr{Cvoid}, Cdouble, Cint, Cint)", "(Ptr{Cvoid}, Cdouble, Cint)").replace(", skip)", ")"), "argument types, the header has 4"),
                      (ok.replace("Cdouble", "Cfloat"), "argument 2 is Cfloat"),
                      (ok.replace("Cint, (Ptr", "Clong, (Ptr"), "returns Clong"),
                      (ok.replace(", skip)", ")"), "3 values for 4 argument types"),
                      (ok.replace("bhip_chains_step", "bhip_chains_stepp"), "no such entry point")):
        problems, _, _ = check_shim(src.replace(ok, bad))
        assert len(problems) == 1 and what in problems[0], (what, problems)
    # and a header that gains a parameter makes the untouched shim red
    hdr = open(HEADER).read().replace("int bhip_chains_step(bhip_chains *ch, double rho, int iters, int skip);",
                                      "int bhip_chains_step(bhip_chains *ch, double rho, int iters, int skip, int flags);")
    problems, _, _ = check_shim(None, hdr)
    assert len(problems) == 1 and "bhip_chains_step" in problems[0]
