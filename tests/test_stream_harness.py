"""bwams/stream.py (the three-step pipeline the tests and bench.py drive mem_process_seqs() with) on the CPU, with a stand-in worker:
a step that fails must end the job with its error — the reader's thread may be blocked handing the next chunk over when step 1 fails
(a run that ran out of HBM with three chunks in flight hung there, profiles/r04_notes.md G)."""
import threading
import time

import pytest

from bwams import stream


class _Seqs:
    def __init__(self, n):
        self.n = n

    def drop_sam(self):
        pass

    def take_sam(self):
        return b""


class _Worker:
    def __init__(self, fail_step, fail_at):
        self.fail_step, self.fail_at, self.calls = fail_step, fail_at, {"stage": 0, "process": 0, "collect": 0}

    def set_deferred_collect(self, on):
        pass

    def _step(self, name):
        k = self.calls[name]
        self.calls[name] += 1
        time.sleep(0.002)
        if name == self.fail_step and k == self.fail_at:
            raise RuntimeError(f"{name} failed at chunk {k}")

    def stage(self, opt, s):
        self._step("stage")

    def process(self, opt, done, s):
        self._step("process")

    def collect(self, opt, s):
        self._step("collect")


def _run(worker, n_chunks, overlap=True):
    out = {}

    def go():
        try:
            out["ok"] = stream.run_job(worker, None, lambda i: _Seqs(10), n_chunks, None, overlap=overlap)
        except RuntimeError as e:
            out["err"] = str(e)

    t = threading.Thread(target=go, daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "run_job did not return"
    return out


def test_job_without_failures_counts_its_reads():
    out = _run(_Worker(None, -1), 7)
    assert out["ok"][1] == 70


@pytest.mark.parametrize("step", ["stage", "process", "collect"])
@pytest.mark.parametrize("at", [0, 2, 5])
def test_a_failing_step_ends_the_job_with_its_error(step, at):
    out = _run(_Worker(step, at), 8)
    assert out.get("err") == f"{step} failed at chunk {at}"


def test_no_overlap_form_raises_too():
    out = _run(_Worker("process", 1), 4, overlap=False)
    assert out.get("err") == "process failed at chunk 1"
