"""Results must not depend on the process that computes them (`-m gpu`).

Which form a library product of the prompt pass runs in (whole / column halves / K segments) and which kernel a 33..160-row
decode projection gets used to be timed once per process; two processes could pick different - numerically different -
forms at a near-tie (round 5 VERDICT weak #2 / ADVICE).  The plans are fixed tables now (openpsg_amd/llm.py): two FRESH
processes must produce identical existence logits, first-step LLM logits and greedy tokens, bit for bit, on the same image
(V4:293-312 via HF-LL:243-281 - the reference's arithmetic has no such freedom either)."""
import json
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe():
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("PSG_PLAN", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tests", "_plan_probe.py")], env=env, cwd=REPO,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("PLAN_PROBE ")][-1]
    return json.loads(line[len("PLAN_PROBE "):])


@pytest.mark.gpu
def test_two_fresh_processes_give_identical_logits_and_tokens():
    a, b = _probe(), _probe()
    print(a)
    assert a["fp32s_prompt_rows"] >= 512 and a["w16_streamed"]          # the shapes the plan tables name were exercised
    assert a == b


def test_plan_tables_are_pure_functions_of_the_shape():
    """CPU: the product path never times anything - the same shape gives the same plan, unknown shapes the plain form."""
    import torch
    from openpsg_amd import llm
    os.environ.pop("PSG_PLAN", None)
    w = torch.empty((22016, 12288), dtype=torch.float16, device="meta")
    assert llm._plan_split_mm(960, w) == ("cols", 2) == llm._plan_split_mm(1040, w)
    assert llm._plan_split_mm(64, w) == ("whole",)
    assert llm._plan_split_mm(960, torch.empty((4096, 33024), dtype=torch.float16, device="meta"), k3=True) == ("kseg", 6)
    assert llm._plan_split_mm(960, torch.empty((4096, 33024), dtype=torch.float16, device="meta"), k3=False) == ("whole",)
    assert llm._plan_split_mm(960, torch.empty((1536, 1536), dtype=torch.float16, device="meta")) == ("whole",)
    x = torch.empty((80, 11008), dtype=torch.float16, device="meta")
    assert llm._plan_batch_mm(x, [torch.empty((4096, 11008), dtype=torch.float16, device="meta")]) == ("own", 128, 1)
    assert llm._plan_batch_mm(x[:, :4096], [torch.empty((32000, 4096), dtype=torch.float16, device="meta")]) == ("lib",)
