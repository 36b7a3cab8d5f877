"""The adapter's stand-in interface is pinned to the reference's real one.

tests/adapter/harness.cpp re-declares IModelComponent / message_data by hand because the
reference's headers need Boost, which this image lacks.  Here -- in the build container only, where
/root/reference exists -- every `virtual` declaration of inst/include/imodel_component.hpp:60-172
(and of IVisitable, its base) and every message_data constructor of
inst/include/message_data.hpp:31-46 is parsed out of the reference's headers and must appear, token
for token, in the stand-in: if either side drifts, this fails."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/inst/include"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree is not on this box")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def norm(text):
    """whitespace-insensitive token string"""
    text = re.sub(r"\s+", " ", text).strip()
    return re.sub(r"\s*([(),&*=;{}:])\s*", r"\1", text)


def class_body(text, name):
    m = re.search(r"\b(?:class|struct)\s+%s\b[^;{]*\{" % name, text)
    assert m, name
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def virtual_signatures(body):
    """every `virtual ...` declaration up to its `;` or inline body, minus `inline` and the body"""
    out = []
    for m in re.finditer(r"\bvirtual\b[^;{]*", body):
        sig = m.group(0).replace("inline", " ")
        out.append(norm(sig))
    return out


def harness_text():
    return norm(strip_comments(open(os.path.join(ROOT, "tests", "adapter", "harness.cpp")).read()))


def test_every_virtual_of_the_reference_interface_is_in_the_stand_in():
    ref = strip_comments(open(os.path.join(REF, "imodel_component.hpp")).read())
    vis = strip_comments(open(os.path.join(REF, "ivisitable.hpp")).read())
    sigs = virtual_signatures(class_body(ref, "IModelComponent")) + \
        [s for s in virtual_signatures(class_body(vis, "IVisitable")) if "~" not in s]
    sigs = [s for s in sigs if "~IModelComponent" not in s]
    # the reference's interface as of v3.5.0: ten virtuals + accept (a new one must be adapted too)
    assert len(sigs) == 11, sigs
    have = harness_text()
    stand_in = norm(class_body(strip_comments(open(os.path.join(ROOT, "tests", "adapter", "harness.cpp")).read()),
                               "IModelComponent"))
    for s in sigs:
        assert s in stand_in, "the reference declares `%s`; tests/adapter/harness.cpp does not" % s
    # ... and the stand-in declares nothing the reference does not (destructor aside)
    for s in virtual_signatures(class_body(strip_comments(open(os.path.join(ROOT, "tests", "adapter", "harness.cpp")).read()),
                                           "IModelComponent")):
        if "~" in s:
            continue
        assert s in sigs, "stand-in only: " + s
    # private getData: the adapter's override must sit behind `private:` like the reference's
    assert re.search(r"private:virtual unitval getData\(", stand_in)
    assert "class IModelComponent:public IVisitable" in norm(ref) or "IVisitable" in ref
    assert have


def test_message_data_constructors_match_the_reference():
    ref = class_body(strip_comments(open(os.path.join(REF, "message_data.hpp")).read()), "message_data")
    ctors = [norm(m.group(0)) for m in re.finditer(r"\bmessage_data\s*\([^)]*\)", ref)]
    assert len(ctors) == 5, ctors
    mine = norm(class_body(strip_comments(open(os.path.join(ROOT, "tests", "adapter", "harness.cpp")).read()),
                           "message_data"))
    for c in ctors:
        assert c in mine, "the reference constructs `%s`; the stand-in does not" % c


def test_the_product_adapter_overrides_exactly_those_virtuals():
    """include/hector_amd_component.hpp marks each of them `override` (so a drift of the base is a
    compile error in a host that builds against the real header)."""
    hdr = strip_comments(open(os.path.join(ROOT, "include", "hector_amd_component.hpp")).read())
    ref = strip_comments(open(os.path.join(REF, "imodel_component.hpp")).read())
    names = set(re.findall(r"\bvirtual\s+[\w:<> ]+?[ &*](\w+)\s*\(", class_body(ref, "IModelComponent")))
    names |= {"accept"}
    names -= {"run_spinup"}   # has a default body in the reference; the adapter keeps it
    for n in sorted(names):
        assert re.search(r"\b%s\s*\([^;{]*\)\s*(const\s*)?override" % n, hdr), n
