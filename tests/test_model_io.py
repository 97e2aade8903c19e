"""Model parameter files and chemistry lookup (L1; docs/faq/chemistry.md:27-56, docs/changelog.md:66): host logic, no GPU."""
import os

import pytest

from ccs_amd import api


def test_json_roundtrip_is_exact(built):
    m = api.default_model()
    t = api.model_to_json(m, ("101-789-500", "101-826-100", "5.0"))
    assert '"ConsensusModelVersion": "ccsx-1"' in t and '"ChemistryName": "SYN-1"' in t
    assert bytes(api.model_from_json(t)) == bytes(m)
    # a perturbed set with awkward floats survives file -> blob -> file unchanged
    api.set_model_name(m, "X-1"); m.snr_lo = 3.3333333; m.trans_poly[5][2][3] = 1.0 / 3.0; m.em_stick[9][1] = 0.25 + 1e-8
    t2 = api.model_to_json(m)
    m2 = api.model_from_json(t2)
    assert bytes(m2) == bytes(m) and api.model_to_json(m2) == t2


def test_bad_files_are_reported(built, tmp_path):
    good = api.model_to_json(api.default_model())
    for text, msg in (("{ nope", "model json"), (good.replace("ccsx-1", "other"), "ConsensusModelVersion"),
                      (good.replace('"SnrRange": [4, 20]', '"SnrRange": [4]'), "SnrRange"),
                      (good.replace("0.295499980", "0.9", 1) if "0.295499980" in good else good.replace('"EmissionMatch": [\n    [', '"EmissionMatch": [\n    [0.5, ', 1), "")):
        with pytest.raises(RuntimeError) as e:
            api.model_from_json(text)
        assert msg in str(e.value)
    with pytest.raises(RuntimeError):
        api.model_load(str(tmp_path / "missing.json"))


def test_chemistry_lookup(built, tmp_path, monkeypatch):
    monkeypatch.delenv("SMRT_CHEMISTRY_BUNDLE_DIR", raising=False)
    m = api.model_for_chemistry("101-789-500", "101-826-100", "5.0.0.6235")      # basecaller versions match on major.minor
    assert m.name == b"SYN-1"
    with pytest.raises(RuntimeError) as e:
        api.model_for_chemistry("101-789-500", "101-826-100", "4.1.0")
    assert "Unsupported chemistries found: (101-789-500/101-826-100/4.1.0)" in str(e.value)
    # injected models: every json under $SMRT_CHEMISTRY_BUNDLE_DIR/arrow/, before the built-in set; broken files are skipped
    (tmp_path / "arrow").mkdir()
    other = api.default_model(); api.set_model_name(other, "EARLY-ACCESS"); other.snr_hi = 25.0
    (tmp_path / "arrow" / "a_broken.json").write_text("[1, 2")
    (tmp_path / "arrow" / "new.json").write_text(api.model_to_json(other, ("102-000-000", "102-111-111", "6.0")))
    monkeypatch.setenv("SMRT_CHEMISTRY_BUNDLE_DIR", str(tmp_path))
    got = api.model_for_chemistry("102-000-000", "102-111-111", "6.0.1")
    assert got.name == b"EARLY-ACCESS" and got.snr_hi == 25.0
    assert api.model_for_chemistry("101-789-500", "101-826-100", "5.0").name == b"SYN-1"


def test_model_json_is_validated_and_escaped(built):
    """every table entry must be finite and a probability where it is one; names / kits are JSON-escaped"""
    import json
    m = api.default_model()
    api.set_model_name(m, 'we"ird\\name')
    text = api.model_to_json(m, ('kit"1', "s\\k", "5.0"))
    d = json.loads(text)                                          # valid JSON despite the quotes / backslashes
    assert d["ChemistryName"] == 'we"ird\\name' and d["Chemistries"][0]["BindingKit"] == 'kit"1'
    back = api.model_from_json(text)
    assert bytes(back) == bytes(m)
    # control characters are written as \\u00XX and the bundled parser reads them back (file -> blob -> file stays the identity)
    api.set_model_name(m, "tab\there\x01")
    text2 = api.model_to_json(m, ("k\n1", "s", "5.0"))
    assert json.loads(text2)["ChemistryName"] == "tab\there\x01" and "\\u0009" in text2
    assert bytes(api.model_from_json(text2)) == bytes(m)
    with pytest.raises(RuntimeError, match="model json"):
        api.model_from_json(text2.replace("\\u0009", "\\u00e9"))
    for key, bad in (("EmissionStick", -0.25), ("EmissionBranch", float("inf")), ("EmissionMatch", 0.0), ("TransitionPolynomials", float("nan"))):
        d2 = json.loads(text)
        row = d2[key][3]
        while isinstance(row[0], list): row = row[0]
        row[1] = bad
        with pytest.raises(RuntimeError, match="model json"):
            api.model_from_json(json.dumps(d2).replace("NaN", "1e999").replace("Infinity", "1e999"))
