"""Host-only: TargetConfig(input) of the C++ front (src/TargetConfig.cpp:120-265 restated) and the JSON reader under
it, through tests/cpp/config_test.  Parity unpinned (the reference holds no test for it and pbcopper's JSON is not
vendored): the cases below follow the statements of TargetConfig.cpp -- required fields and their error texts, the
"M184VI" position syntax, the wrapping characters the CLI may leave around the argument, file vs string input."""
import json
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(HERE, "cpp", "config_test")

CONFIG = {
    "referenceName": "toy", "version": "v1", "databaseVersion": "db 1.0", "referenceSequence": "ACGTACGTAC",
    "genes": [
        {"name": "Reverse Transcriptase", "begin": 2550, "end": 4229,
         "drms": [{"name": "NRTI", "positions": ["M184VI", "K65R", "T69", "151M", "x-Y115F!"]},
                  {"name": "NNRTI", "positions": ["KR103NS"]}],
         "expectedminors": [{"position": 184, "aminoacid": "V", "codon": "GTG"}]},
        {"name": "Protease", "begin": 2253, "end": 2549},
    ],
}


def run(*args):
    assert os.path.exists(BIN), "tests/cpp/config_test not built: run __graft_entry__.build()"
    out = subprocess.run([BIN, *args], capture_output=True, encoding="utf-8", timeout=60)
    assert out.returncode == 0, out.stderr
    return out.stdout.splitlines(), out.stderr


def test_config_from_string_and_file(tmp_path):
    text = json.dumps(CONFIG)
    path = str(tmp_path / "config.json")
    open(path, "w").write(json.dumps(CONFIG, indent=2))
    want = [
        "CONFIG name=toy version=v1 db=db 1.0 ref=ACGTACGTAC genes=2 minors=1",
        "GENE Reverse Transcriptase 2550 4229",
        "DRM NRTI M184V M184I K65R T69* *151M Y115F",  # every (ref, variant) pair; a missing side is '*'; first match only
        "DRM NNRTI K103N K103S R103N R103S",
        "MINOR 184 V GTG",
        "GENE Protease 2253 2549",
    ]
    for arg in (text, "<" + text + ">", "'" + text + "'", '"' + text + '"', path, "<" + path + ">"):
        lines, _ = run(arg)
        assert lines == want, arg[:40]


def test_config_errors_and_empty_inputs(tmp_path):
    def err(cfg):
        lines, _ = run(json.dumps(cfg))
        assert len(lines) == 1 and lines[0].startswith("ERR "), lines
        return lines[0][4:]

    assert err({"genes": [{"begin": 1, "end": 9}]}) == "Missing name in gene."
    assert err({"genes": [{"name": "g", "end": 9}]}) == "Missing begin in gene."
    assert err({"genes": [{"name": "g", "begin": 1}]}) == "Missing end in gene."
    assert err({"genes": [{"name": "g", "begin": 1, "end": 9, "drms": [{"positions": ["K65R"]}]}]}) == "Missing name in drm in gene g"
    assert err({"genes": [{"name": "g", "begin": 1, "end": 9, "drms": [{"name": "d"}]}]}) == "Missing positions in drm in gene g"
    assert err({"genes": [{"name": "g", "begin": 1, "end": 9, "drms": [{"name": "d", "positions": []}]}]}) == "Empty positions in drm in gene g"
    for missing, text in (("position", "Missing position in expectedminors in gene g"), ("aminoacid", "Missing aminoacid in expectedminors in gene g"),
                          ("codon", "Missing codon in expectedminors in gene g")):
        minor = {"position": 3, "aminoacid": "K", "codon": "AAA"}
        del minor[missing]
        assert err({"genes": [{"name": "g", "begin": 1, "end": 9, "expectedminors": [minor]}]}) == text
    # no genes, empty object, empty string: an empty configuration (the caller then treats the region as one ORF)
    for arg in ("{}", '{"genes": []}', '{"referenceName": "x"}'):
        lines, _ = run(arg)
        assert lines[0].endswith("genes=0 minors=0")
    lines, stderr = run("HIV")  # a built-in tag of the reference: not shipped here, reported like any unknown input
    assert lines == ["CONFIG name= version= db= ref= genes=0 minors=0"]
    assert "Warning: Target config -c is not a file, JSON string, or tag: HIV" in stderr
    lines, _ = run("{ not json")
    assert lines[0].startswith("ERR JSON parse error")


@pytest.mark.parametrize("text", [
    '{"a": [1, -2, 3.5, 1e3, true, false, null], "b": {"c": "d\\n\\u00e9\\ud83d\\ude00"}, "e": ""}',
    "[]", "{}", '"x"', "0", "-0.0", "18446744073709551615", "[[[[1]]]]", ' {"k" : [ ] } ',
])
def test_json_reader_round_trip(text):
    lines, _ = run("--json", text)
    assert lines[0].startswith("JSON ")
    assert json.loads(lines[0][5:]) == json.loads(text)


@pytest.mark.parametrize("text", ["", "{", "[1,]", '{"a":}', '{"a" 1}', "01", "1.", "tru", '"\\x"', '"abc', "[1] 2", '{"a":1,}', '"\\ud800"'])
def test_json_reader_rejects(text):
    lines, _ = run("--json", text)
    assert lines[0].startswith("ERR JSON parse error"), lines
