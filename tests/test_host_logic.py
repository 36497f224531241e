"""CPU-only tests: host glue of the drop-in (validation, preparation), the C-ABI library's exported
symbols, synthetic generators, and that nothing in the product imports the oracle."""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest

import goldens
from oracle import disort_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pythonic-disort_amd", "pydisort_amd")


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rtd.h")).read()
    declared = set(re.findall(r"\b(rtd_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18
    from pydisort_amd import _lib
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load().rtd_version() >= 100


def test_stub_transport_is_test_infrastructure_with_the_entry_points_librtd_binds():
    """tests/stub/librccl_stub.so (the stand-in that lets rank > 0 of the data plane execute on one GPU): exports exactly what
    RcclApi binds plus the marker rtd_comm_transport asks for; the product never names it -- it is reached only through the
    RTD_RCCL_STUB environment variable, which nothing under pythonic-disort_amd/ sets."""
    stub = os.path.join(ROOT, "tests", "stub", "librccl_stub.so")
    assert os.path.exists(stub), "python tests/stub/build_stub.py (done by __graft_entry__.build())"
    lib = ctypes.CDLL(stub)
    api = open(os.path.join(ROOT, "pythonic-disort_amd", "csrc", "rtd_api.hip")).read()
    bound = set(re.findall(r'dlsym\(api\.h, "([A-Za-z]+)"\)', api))
    assert len(bound) == 14 and "rcclStubTransport" in bound
    for name in bound:
        assert hasattr(lib, name), name
    uid = ctypes.create_string_buffer(128)
    assert lib.ncclGetUniqueId(uid) == 0 and uid.value.startswith(b"/rccl_stub_")  # (no HIP call: works without a GPU)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pythonic-disort_amd")):
        for f in files:
            if f.endswith(".py"):
                assert "RTD_RCCL_STUB" not in open(os.path.join(dirpath, f)).read(), f
    assert "RTD_RCCL_STUB" not in open(os.path.join(ROOT, "__graft_entry__.py")).read()


def test_header_is_plain_c_and_the_c_example_links():
    """include/rtd.h is the drop-in boundary: plain C (C99, -pedantic -Werror), usable without C++ or Python -- and
    examples/solve_columns.c, a whole solve driven from C, compiles against it and links to librtd.so (it runs on the GPU box:
    tests/test_gpu_c_example.py)."""
    import shutil
    import subprocess
    import tempfile
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    libdir = os.path.join(ROOT, "pythonic-disort_amd", "pydisort_amd")
    with tempfile.TemporaryDirectory() as d:
        with open(os.path.join(d, "t.c"), "w") as f:
            f.write('#include "rtd.h"\nint main(void) { rtd_dims d; rtd_inputs in; (void)d; (void)in; return rtd_version() > 0 ? 0 : 1; }\n')
        subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        os.path.join(d, "t.c")], check=True)
        subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "solve_columns.c"), os.path.join(libdir, "librtd.so"), "-lm", "-Wl,-rpath," + libdir,
                        "-o", os.path.join(d, "solve_columns")], check=True)


def test_product_never_imports_oracle_or_reference():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pythonic-disort_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "PythonicDISORT" not in src or "import PythonicDISORT" not in src, f
                assert "/root/reference" not in src, f


def test_library_reads_only_the_documented_environment():
    """The release library must not carry result-changing switches behind environment variables (round-3 verdict: RTD_BC_ALIAS):
    every getenv of the sources is on the list of include/rtd.h, and that list names nothing the sources do not read."""
    hdr = open(os.path.join(ROOT, "include", "rtd.h")).read()
    block = hdr[hdr.index("environment read by the library"):]
    documented = set(re.findall(r"^ \*   (RTD_[A-Z0-9_]+) ", block, flags=re.M))
    assert documented, "include/rtd.h lists no environment variables"
    read = set()
    csrc = os.path.join(ROOT, "pythonic-disort_amd", "csrc")
    for f in os.listdir(csrc):
        src = open(os.path.join(csrc, f)).read()
        calls = re.findall(r"getenv\(([^)]*)\)", src)
        for arg in calls:
            m = re.fullmatch(r'\s*"([A-Za-z0-9_]+)"\s*', arg)
            assert m, f"{f}: getenv of a computed name: {arg}"
            read.add(m.group(1))
    assert read == documented, (read ^ documented)
    for name in read:  # what a name promises: an implementation choice, a buffer size or diagnostics -- never an experiment
        assert not re.search(r"ALIAS|EXPERIMENT|GARBAGE", name), name
    # the Python side: RTD_LIB (which build of the same ABI to load) and the bench / test harness variables only
    for f in os.listdir(PKG):
        if f.endswith(".py"):
            for name in re.findall(r"environ(?:\.get)?\W+(RTD_[A-Z0-9_]+)", open(os.path.join(PKG, f)).read()):
                assert name in {"RTD_LIB", "RTD_CTL_DIR", "RTD_CTL_TIMEOUT"}, (f, name)  # (_control.py: where ranks meet, how long they wait)


def _kw(tid="9c"):
    return goldens.load(tid)[0]["kwargs"]


@pytest.mark.parametrize("tid", ["1a", "3a", "5b", "7c", "8ARTS_B", "9c", "11a", "Ic"])
def test_prepare_matches_oracle_prepare(tid):
    """Host preparation (delta-M scaling, source rescale) against the oracle's restatement."""
    from pydisort_amd._prepare import prepare_columns
    for call in goldens.load(tid):
        kw = call["kwargs"]
        p = O.prepare(**kw)
        L, N, M = p["L"], p["N"], p["M"]
        Leg = np.atleast_2d(kw["Leg_coeffs_all"]).astype(float)
        Leg[:, 0] = 1.0
        s = np.atleast_2d(kw["s_poly_coeffs"])
        s = s[:, :p["Ns"]] if p["Ns"] else np.zeros((L, 0))
        f = np.broadcast_to(np.atleast_1d(kw["f_arr"]), (L,))
        bq = np.stack([t[0] for t in p["bdrf"]]) if p["bdrf"] else np.zeros((0, N, N))
        bq0 = np.stack([t[1] for t in p["bdrf"]]) if p["bdrf"] else np.zeros((0, N))

        def bc(b):
            out = np.zeros((N, M))
            b = np.asarray(b, float)
            if b.size == 1:
                out[:, 0] = b.reshape(-1)[0]
            elif b.ndim == 1:
                out[:, 0] = b
            else:
                out[:] = b
            return out
        q = prepare_columns(np.atleast_1d(kw["tau_arr"])[None], np.atleast_1d(kw["omega_arr"])[None], kw["NQuad"],
                            Leg[None], [kw["mu0"]], [kw["I0"]], [kw["phi0"]], p["P"], M, bc(kw["b_pos"])[None],
                            bc(kw["b_neg"])[None], f[None], s[None], bq[None], bq0[None])
        for a, b in [("omega_s", "omega_s"), ("tau_s0", "tau_s0"), ("scale_tau", "scale_tau"), ("wleg", "wleg")]:
            assert np.allclose(q[a][0], p[b], rtol=1e-14, atol=1e-15), a
        assert np.isclose(q["rescale"][0], p["rescale"], rtol=1e-14)
        assert np.isclose(q["I0"][0], p["I0"], rtol=1e-14)
        assert np.allclose(q["b_pos"][0].T, p["b_pos"], rtol=1e-14) and np.allclose(q["b_neg"][0].T, p["b_neg"], rtol=1e-14)
        if p["Ns"]:
            assert np.allclose(q["s_s"][0], p["s_s"], rtol=1e-13, atol=1e-15)


BAD = [
    (dict(tau_arr=-1.0), "tau values cannot be non-positive"),
    (dict(tau_arr=np.array([1.0, 0.5]), omega_arr=np.array([0.5, 0.5]), Leg_coeffs_all=np.ones((2, 9)) * 0.5), "thicknesses"),
    (dict(omega_arr=1.0), "Single-scattering albedo"),
    (dict(NLeg=100), "`NLeg` cannot be larger"),
    (dict(NQuad=7), "even"),
    (dict(NFourier=0), "Fourier modes to use in the solution must be positive"),
    (dict(NFourier=9), "less than or equal to the number of phase function"),
    (dict(I0=-1.0), "cannot be negative"),
    (dict(mu0=1.5), "cosine of the polar angle"),
    (dict(phi0=7.0), "principal azimuthal angle"),
    (dict(b_pos=np.ones(3)), "bottom boundary condition"),
    (dict(b_neg=np.ones((2, 2))), "top boundary condition"),
    (dict(f_arr=1.5), "fractional scattering"),
    (dict(use_banded_solver_NLayers=2), "minimum threshold"),
]


@pytest.mark.parametrize("override,msg", BAD)
def test_input_checks_raise_like_the_reference(override, msg):
    """Same hard-error conditions as pydisort.py:222-291 (checked before any device work)."""
    import pydisort_amd
    kw = dict(tau_arr=1.0, omega_arr=0.5, NQuad=8, Leg_coeffs_all=np.array([1.0, 0.5, 0.2, 0.1, 0.05, 0.02, 0.01, 0.005, 0.001]),
              mu0=0.5, I0=1.0, phi0=0.0)
    kw.update(override)
    with pytest.raises(ValueError, match=re.escape(msg)):
        pydisort_amd.pydisort(**kw)


def _frontend_error_cases():
    import json
    import frontend_mutations as F
    table = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frontend_errors.json")))
    return [(name, seed, table[f"{name}|{seed}"]) for name in F.MUTATIONS for seed in F.SEEDS]


@pytest.mark.parametrize("name,seed,want", _frontend_error_cases())
def test_invalid_inputs_raise_what_the_reference_raises(name, seed, want):
    """31 kinds of invalid input x 3 atmospheres: the exception type AND text of the reference (recorded from the reference
    itself by tests/golden/make_frontend_error_goldens.py; pydisort.py:222-291), raised before any device work.  Inputs the
    reference accepts (negative boundary sources, a zeroth moment other than 1 ...) must pass the drop-in's checks too: what
    then fails here, without a GPU, is the device layer, not a ValueError."""
    import frontend_mutations as F
    import pydisort_amd
    kw = F.case(name, seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        try:
            pydisort_amd.pydisort(**kw)
            got = ["accepted", ""]
        except Exception as e:  # noqa: BLE001
            got = [type(e).__name__, str(e)]
    if want[0] == "accepted":
        assert got[0] != "ValueError", got
    else:
        assert got == want


def test_synthetic_columns_are_deterministic_and_independent_of_batch():
    from pydisort_amd import synthetic
    a = synthetic.cfg4_columns(8)
    b = synthetic.cfg4_columns(3, first=5)
    for k in ("tau_arr", "omega_arr", "Leg_coeffs_all", "mu0", "f_arr"):
        assert np.array_equal(a[k][5:8], b[k])
    c5 = synthetic.cfg5_columns(2)
    assert c5["bdrf_q"].shape == (2, 2, 32, 32) and c5["Leg_coeffs_all"].shape == (2, 50, 65)


def test_oracle_on_synthetic_config_is_self_consistent():
    """cfg4 column through the oracle: banded (L >= 10) and dense BC solvers agree."""
    from pydisort_amd import synthetic
    cfg = synthetic.cfg4_columns(1, L=12, NQuad=8)
    kw = synthetic.column_kwargs(cfg, 0)
    a = O.pydisort(**kw)
    b = O.pydisort(use_banded_solver_NLayers=100, **kw)
    tau = np.linspace(0, kw["tau_arr"][-1], 7)
    assert np.allclose(a[4](tau, 0.3), b[4](tau, 0.3), rtol=1e-10)


def test_pair_layout_jacobi_schedule_meets_every_pair_once():
    """The butterfly ordering of the device's Jacobi sweeps (csrc/rtd_eig.hip: JSched), replayed on the host."""
    import importlib.util
    import random
    spec = importlib.util.spec_from_file_location("jacobi_schedule", os.path.join(ROOT, "tools", "jacobi_schedule.py"))
    js = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(js)
    rng = random.Random(7)
    for NP in (4, 8, 16, 32):
        sw, mk = js.build(NP)
        assert len(sw) == NP - 1 and set(mk) <= js.DPP_XOR_MASKS
        js.replay(NP, 3)
        perm = list(range(NP))
        rng.shuffle(perm)
        js.replay(NP, 3, (perm[: NP // 2], perm[NP // 2:]))


def test_assemble_shim_has_the_references_signature():
    """pydisort_amd._assemble._assemble_intensity_and_fluxes takes the reference's 34 parameters, same names, same order
    (_assemble_intensity_and_fluxes.py:8-32; the names were recorded from the reference next to the captured arguments)."""
    import inspect
    import os
    import numpy as np
    from pydisort_amd._assemble import _assemble_intensity_and_fluxes as shim
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "assemble")
    z = np.load(os.path.join(d, "1a.npz"))
    ref_names = [str(s) for s in z["names"]]
    assert len(ref_names) == 34
    params = list(inspect.signature(shim).parameters)
    assert params[:34] == ref_names and params[34:] == ["device"]
    assert len([f for f in os.listdir(d) if f.endswith(".npz")]) >= 8


def test_double_double_taylor_shift_is_exact_to_the_last_bit():
    """csrc/rtd_dd.h moves the origin of the thermal source polynomials to the top of their layer (one double-double Taylor shift
    on upload instead of a cancelling float64 evaluation at every use, subroutines.py:746-862 / pydisort.py:316-338).  Compiled
    for the CPU here (the header is host + device) and held to exact rational arithmetic: every shifted coefficient is the
    correctly rounded value, also where the terms cancel over ten digits (8ARTS_A's bottom layers: a0 = -3.6e5, a1 tau = +3.6e5)."""
    import shutil
    import subprocess
    import tempfile
    from fractions import Fraction
    from math import comb
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    rng = np.random.default_rng(3)
    cases = [(2, 21.99325904, [-3.61508536e+05, 1.63496454e+04]), (2, 21.9931618, [-5.36654739e+04, 2.42711395e+03]),
             (1, 5.0, [1.25]), (3, 0.0, [1.0, -2.0, 0.5])]
    for _ in range(200):
        n = int(rng.integers(1, 7))
        t = float(rng.uniform(0.0, 60.0))
        c = (rng.normal(size=n) * 10.0 ** rng.uniform(-3, 6, size=n)).tolist()
        if n >= 2 and rng.random() < 0.5:  # make the constant term cancel against the rest at t
            c[0] = -sum(c[j] * t**j for j in range(1, n)) * (1.0 + 1e-9 * rng.normal())
        cases.append((n, t, c))
    # round-5 advice: the work array holds 16 double-doubles and a 17th coefficient ran past it (the device-prepare path had no
    # cap).  16 is the last double-double length; 17 and more -- the reference puts no limit on Nscoeffs -- take the float64 scheme
    # in place.  Built with the address sanitizer: an overrun of the stack array ends the harness.
    for n in (15, 16, 17, 24):
        for _ in range(3):
            cases.append((n, float(rng.uniform(0.0, 1.5)), rng.normal(size=n).tolist()))
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "dd_shift")
        subprocess.run([gxx, "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        os.path.join(ROOT, "tests", "cpu", "dd_shift.cpp"), "-o", exe], check=True)
        text = "".join(f"{n} {t!r} " + " ".join(repr(float(x)) for x in c) + "\n" for n, t, c in cases)
        out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == len(cases)
    for (n, t, c), line in zip(cases, out):
        got = [float(x) for x in line.split()]
        tf, cf = Fraction(t), [Fraction(float(x)) for x in c]
        for i in range(n):
            exact = sum(cf[j] * comb(j, i) * tf ** (j - i) for j in range(i, n))
            want = float(exact)  # correctly rounded
            if n <= 16:
                assert got[i] == want or abs(got[i] - want) <= abs(want) * 2.3e-16, (n, t, c, i, got[i], want)
            else:  # float64 Horner: a backward-stable evaluation, measured against the size of the terms it adds
                size = float(sum(abs(cf[j]) * comb(j, i) * tf ** (j - i) for j in range(i, n)))
                assert abs(got[i] - want) <= 4e-15 * size, (n, t, i, got[i], want)
