"""Boundary conformance against the reference's REAL headers.

barlib cannot be compiled in this image (fftw3.h / gsl are absent), so the C++ host layer is compiled against
``HamilView`` / ``HamilNumericalView`` and the barlib-side shim in INTEGRATION.md is a listing.  This test reads
``/root/reference/barlib/include/struct_hamil.h``, ``struct_main.h``, ``curses_funcs.h`` and ``src/HMC.cc`` as TEXT (no
code from the tree is imported, built or copied; nothing travels to the GPU box, where the test is skipped) and asserts

  * every member of HamilNumericalView / HamilView that claims to be a HAMIL_NUMERICAL / HAMIL_DATA member exists
    upstream under that name with that type (input pointers may add ``const``); members the shim adds are listed here
    explicitly, so a new one cannot slip in unnoticed;
  * every ``n->x`` / ``hd->x`` / ``data->numerical->x`` / ``data->curses->x`` the INTEGRATION.md listing touches exists in
    HAMIL_NUMERICAL / HAMIL_DATA / NUMERICAL / CURSES_STRUCT, every ``c.x`` it assigns is a bchmc_config field and every
    ``bchmc_*`` it calls is declared in include/bchmc.h;
  * the four function signatures the listing replaces equal the ones in HMC.cc (return type, name, parameter types);
  * the step-size keys of EpsAdaptConfig are NUMERICAL / HAMIL_NUMERICAL members of the same type.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/barlib"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is only present in the build container")


def _strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def _struct_body(text, name):
    """Text between the braces of `struct <name> {`, nested blocks (constructors, nested structs) removed."""
    m = re.search(r"\bstruct\s+%s\b\s*\{" % re.escape(name), text)
    assert m, "struct %s not found" % name
    depth, i, out = 1, m.end(), []
    while depth:
        ch = text[i]
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth == 1:
                out.append(" __BLOCK__ ")  # a nested block (constructor body, nested struct) collapsed
        elif depth == 1:
            out.append(ch)
        i += 1
    return "".join(out)


def _members(body):
    """{name: normalised type} of the plain data members of a struct body (function pointers, constructors and nested
    types are skipped)."""
    out = {}
    for stmt in body.split(";"):
        stmt = " ".join(stmt.split())
        nested = re.match(r"^struct (\w+) __BLOCK__ (\w+)$", stmt)  # struct T { ... } name;
        if nested:
            out[nested.group(2)] = nested.group(1)
            continue
        if not stmt or "(" in stmt or "__BLOCK__" in stmt or stmt.startswith(("struct EpsAdapt", "using ", "typedef ")):
            continue
        stmt = re.sub(r"=[^,]*", "", stmt)            # default member initialisers
        stmt = re.sub(r"\[[^\]]*\]", "", stmt)        # array extents
        stmt = " ".join(stmt.split())
        m = re.match(r"^((?:const\s+)?(?:struct\s+)?[\w:<> ]+?)\s*((?:\*\s*)?\w+(?:\s*,\s*(?:\*\s*)?\w+)*)$", stmt)
        if not m:
            continue
        base = m.group(1).strip()
        for decl in m.group(2).split(","):
            decl = decl.strip()
            stars = decl.count("*")
            out[decl.replace("*", "").strip()] = _norm_type(base + " *" * stars)
    return out


def _norm_type(t):
    t = t.replace("struct ", "").replace("std::", "")
    t = re.sub(r"\bunsigned int\b", "unsigned", t)
    t = re.sub(r"\buint64_t\b", "uint64_t", t)
    return " ".join(t.replace("*", " * ").split())


def _read(*parts):
    with open(os.path.join(*parts)) as f:
        return f.read()


@pytest.fixture(scope="module")
def ref():
    hamil = _strip_comments(_read(REF, "include", "struct_hamil.h"))
    main = _strip_comments(_read(REF, "include", "struct_main.h"))
    curses = _strip_comments(_read(REF, "include", "curses_funcs.h"))
    return dict(HAMIL_NUMERICAL=_members(_struct_body(hamil, "HAMIL_NUMERICAL")),
                HAMIL_DATA=_members(_struct_body(hamil, "HAMIL_DATA")),
                NUMERICAL=_members(_struct_body(main, "NUMERICAL")),
                OBSERVATIONAL=_members(_struct_body(main, "OBSERVATIONAL")),
                DATA=_members(_struct_body(main, "DATA")),
                CURSES_STRUCT=_members(_struct_body(curses, "CURSES_STRUCT")),
                hmc=_strip_comments(_read(REF, "src", "HMC.cc")))


@pytest.fixture(scope="module")
def ours():
    hpp = _strip_comments(_read(ROOT, "include", "bchmc_shim.hpp"))
    abi = _strip_comments(_read(ROOT, "include", "bchmc.h"))
    md = _read(ROOT, "INTEGRATION.md")
    listing = re.search(r"```cpp\n#include \"struct_main.h\"(.*?)```", md, flags=re.S)
    assert listing, "INTEGRATION.md lost its shim listing"
    return dict(numerical=_members(_struct_body(hpp, "HamilNumericalView")),
                view=_members(_struct_body(hpp, "HamilView")),
                eps_cfg=_members(_struct_body(hpp, "EpsAdaptConfig")),
                config=_members(_struct_body(abi, "bchmc_config")),
                abi=abi, listing=_strip_comments(listing.group(1)))


# Members the shim ADDS to its view of HAMIL_DATA (no upstream counterpart); everything else must exist upstream.
SHIM_ONLY_VIEW = {
    "likelihood",          # which plugin functions set_likelihood_functions bound: DATA::numerical->likelihood upstream
    "device", "engine", "eps", "comm", "comm_rank", "inputs_generation", "uploaded_generation", "deterministic",
    "mass_generation", "mass_uploaded_generation", "reuse_eom_energies", "eom",
}


def _same_type(ours_t, ref_t, allow_const=False):
    if allow_const:
        ours_t = ours_t.replace("const ", "")
    return ours_t == ref_t


def test_parser_sees_the_reference_structs(ref):
    # anchors: a few members whose declarations exercise the parser's cases (multi-declarator line, pointer, bool)
    hn, hd = ref["HAMIL_NUMERICAL"], ref["HAMIL_DATA"]
    assert hn["N1"] == "unsigned" and hn["N"] == "ULONG" and hn["planepar"] == "bool"
    assert hn["psi_likeli_f"] == "real_prec" and hn["H_kin_f"] == "real_prec" and hn["dH"] == "real_prec"
    assert hd["numerical"] == "HAMIL_NUMERICAL *" and hd["signal_PS"] == "real_prec *" and hd["rsd_model"] == "bool"
    assert ref["NUMERICAL"]["acc_flag_N_a"] == "vector<bool>" and ref["NUMERICAL"]["epsilon_N_a"] == "vector<real_prec>"
    assert ref["CURSES_STRUCT"]["table"] == "WINDOW *" and ref["DATA"]["curses"] == "CURSES_STRUCT *"
    assert len(hn) > 70 and len(hd) > 35


def test_numerical_view_members_are_hamil_numerical_members(ref, ours):
    hn = ref["HAMIL_NUMERICAL"]
    assert len(ours["numerical"]) >= 40
    for name, typ in ours["numerical"].items():
        assert name in hn, "HamilNumericalView::%s is not a member of HAMIL_NUMERICAL" % name
        assert _same_type(typ, hn[name]), "HamilNumericalView::%s is %s, HAMIL_NUMERICAL::%s is %s" % (name, typ, name, hn[name])
    # cubic grids: the view keeps N1 / L1 only, the engine requires the other axes to be equal upstream (init_par.cc:116-118)
    for name in ("N2", "N3", "L2", "L3"):
        assert name in hn


def test_view_members_are_hamil_data_members_or_declared_additions(ref, ours):
    hd = ref["HAMIL_DATA"]
    seen_additions = set()
    for name, typ in ours["view"].items():
        if name in SHIM_ONLY_VIEW:
            seen_additions.add(name)
            assert name not in hd, "%s is listed as a shim addition but exists upstream" % name
            continue
        assert name in hd, "HamilView::%s is neither a HAMIL_DATA member nor a declared shim addition" % name
        if name == "numerical":
            assert typ == "HamilNumericalView *" and hd[name] == "HAMIL_NUMERICAL *"
            continue
        assert _same_type(typ, hd[name], allow_const=True), "HamilView::%s is %s, HAMIL_DATA::%s is %s" % (name, typ, name, hd[name])
    assert seen_additions == SHIM_ONLY_VIEW & set(ours["view"]), "stale entry in SHIM_ONLY_VIEW"
    assert ref["NUMERICAL"]["likelihood"] == "int" == ours["view"]["likelihood"]
    # the five plugin function pointers the likelihood enum stands for exist in both structs (struct_main.h:66-72)
    hamil_txt = _strip_comments(_read(REF, "include", "struct_hamil.h"))
    main_txt = _strip_comments(_read(REF, "include", "struct_main.h"))
    for fn in ("partial_f_delta_x_log_like", "log_like", "grad_f_delta_x_comp", "log_prior", "grad_log_prior"):
        assert re.search(r"\(\s*\*\s*%s\s*\)" % fn, hamil_txt) and re.search(r"\(\s*\*\s*%s\s*\)" % fn, main_txt)


def test_eps_adapt_keys_are_reference_members(ref, ours):
    num, hn = ref["NUMERICAL"], ref["HAMIL_NUMERICAL"]
    for name, typ in ours["eps_cfg"].items():
        src = num if name in num else hn
        assert name in src, "EpsAdaptConfig::%s exists in neither NUMERICAL nor HAMIL_NUMERICAL" % name
        assert _same_type(typ, src[name]), "EpsAdaptConfig::%s is %s upstream %s" % (name, typ, src[name])
    assert num["count_attempts"] == "ULONG"


def _signature(text, name):
    """(return type, [parameter types]) of the definition of `name` in C++ text."""
    m = re.search(r"(?:^|\n)\s*([\w ]+?)\s+%s\s*\(([^)]*)\)\s*\{" % re.escape(name), text)
    assert m, "no definition of %s" % name
    params = []
    for par in m.group(2).split(","):
        par = " ".join(par.replace("*", " * ").split())
        toks = par.split()
        params.append(_norm_type(" ".join(toks[:-1])))  # drop the parameter name
    return _norm_type(m.group(1)), params


@pytest.mark.parametrize("fn", ["Hamiltonian_EoM", "delta_Hamiltonian", "kinetic_term", "psi"])
def test_listing_signatures_equal_the_reference(ref, ours, fn):
    assert _signature(ours["listing"], fn) == _signature(ref["hmc"], fn)


def test_listing_touches_only_existing_members_and_declared_entry_points(ref, ours):
    lst = ours["listing"]
    used = dict(n=set(re.findall(r"\bn->(\w+)", lst)), hd=set(re.findall(r"\bhd->(\w+)", lst)),
                num=set(re.findall(r"\bdata->numerical->(\w+)", lst)),
                cur=set(re.findall(r"\bdata->curses->(\w+)", lst)))
    assert len(used["n"]) > 30 and len(used["hd"]) > 15 and used["num"] and used["cur"]
    for x in used["n"]:
        assert x in ref["HAMIL_NUMERICAL"], "listing uses n->%s, not a HAMIL_NUMERICAL member" % x
    for x in used["hd"]:
        assert x in ref["HAMIL_DATA"], "listing uses hd->%s, not a HAMIL_DATA member" % x
    for x in used["num"]:
        assert x in ref["NUMERICAL"], "listing uses data->numerical->%s, not a NUMERICAL member" % x
    for x in used["cur"]:
        assert x in ref["CURSES_STRUCT"], "listing uses data->curses->%s, not a CURSES_STRUCT member" % x
    # types of what crosses into bchmc_config: integer / bool / real_prec members only
    scalar = {"unsigned", "int", "bool", "real_prec", "ULONG"}
    for x in re.findall(r"c\.\w+\s*=\s*n->(\w+)", lst):
        assert ref["HAMIL_NUMERICAL"][x] in scalar
    for x in re.findall(r"c\.\w+\s*=\s*hd->(\w+)", lst):
        assert ref["HAMIL_DATA"][x] in scalar
    for x in re.findall(r"\bc\.(\w+)\s*=", lst):
        assert x in ours["config"], "listing assigns bchmc_config::%s, which include/bchmc.h does not declare" % x
    for fn in set(re.findall(r"\b(bchmc_\w+)\s*\(", lst)):
        assert re.search(r"\b%s\s*\(" % fn, ours["abi"]), "listing calls %s, which include/bchmc.h does not declare" % fn
    # arrays handed to bchmc_upload / bchmc_fetch are real_prec* members (DOUBLE_PREC: double)
    for x in re.findall(r"bchmc_(?:upload|fetch)\([^;]*?hd->(\w+)", lst):
        assert ref["HAMIL_DATA"][x] == "real_prec *"


def test_call_order_the_shim_relies_on(ref):
    """HamiltonianMC calls Hamiltonian_EoM and delta_Hamiltonian back to back on the same four arrays (HMC.cc:455-459):
    the contract behind bchmc_leapfrog_dh / HamilView::reuse_eom_energies."""
    m = re.search(r"Hamiltonian_EoM\s*\(\s*hd\s*,([^;]*?),\s*seed\s*,\s*data\s*\)\s*;(.*?)delta_Hamiltonian\s*\(\s*hd\s*,([^;]*?),\s*data\s*\)",
                  ref["hmc"], flags=re.S)
    assert m, "HamiltonianMC no longer calls the two functions in sequence"
    norm = lambda s: [a.strip() for a in s.split(",")]  # noqa: E731
    assert norm(m.group(1)) == norm(m.group(3)) and len(norm(m.group(1))) == 4
    between = m.group(2)
    assert not re.search(r"\b(signali|signalf|momentai|momentaf)\b[^;]*=", between), "the arrays are written in between"
