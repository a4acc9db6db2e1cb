"""A job of many chunks through the compiled outer boundary, the way the reference's kt_pipeline runs it
(/root/reference/src/fastmap.cpp:307-468): step 0 reads a chunk, step 1 is mem_process_seqs, step 2 writes — three threads, a step
holds one chunk at a time and chunks keep their order (the ordering lock, :475-491).  With a worker of depth >= 2 the reader stages
chunk i + 1 (records -> page-locked arrays -> the devices) and the writer collects chunk i - 1 (SAM text down) while chunk i computes:
host/mem_process_seqs_hip.cpp.  Test and bench harness; the product is the host layer it drives."""
import queue
import threading
import time

from . import capi


def run_job(worker: "capi.Worker", opt: "capi.MemOptT", make_chunk, n_chunks: int, sink, n_processed0: int = 0, overlap: bool = True):
    """make_chunk(i) -> capi.Seqs (step 0's parsing; called on the reader's thread); sink(i, sam_bytes) (step 2's fputs; sink=None:
    the strings are freed unread, Seqs.drop_sam).
    overlap=False runs the three steps of every chunk one after the other on this thread (the strict drop-in: mem_process_seqs alone).
    Returns (wall seconds, reads)."""
    reads = 0
    t0 = time.perf_counter()
    if not overlap:
        worker.set_deferred_collect(False)
        done = n_processed0
        for i in range(n_chunks):
            s = make_chunk(i)
            worker.process(opt, done, s)
            (s.drop_sam() if sink is None else sink(i, s.take_sam()))
            done += s.n
            reads += s.n
        return time.perf_counter() - t0, reads
    worker.set_deferred_collect(True)
    q01, q12 = queue.Queue(maxsize=1), queue.Queue(maxsize=1)
    errs = []

    def reader():
        try:
            for i in range(n_chunks):
                if errs:                              # a later step has failed: nothing more to stage
                    break
                s = make_chunk(i)
                worker.stage(opt, s)                  # waits for a free slot: at most `depth` chunks in flight
                q01.put((i, s))
        except BaseException as e:                   # noqa: BLE001 - handed to the caller's thread
            errs.append(e)
        q01.put(None)

    def writer():
        try:
            while True:
                it = q12.get()
                if it is None:
                    return
                i, s = it
                worker.collect(opt, s)
                (s.drop_sam() if sink is None else sink(i, s.take_sam()))
        except BaseException as e:                   # noqa: BLE001
            errs.append(e)
            while q12.get() is not None:             # keep draining so that step 1 never blocks on a dead writer
                pass

    tr, tw = threading.Thread(target=reader), threading.Thread(target=writer)
    tr.start(); tw.start()
    done = n_processed0
    try:
        while True:
            it = q01.get()
            if it is None:
                break
            i, s = it
            if not errs:
                try:
                    worker.process(opt, done, s)
                    q12.put((i, s))
                except BaseException as e:           # noqa: BLE001 - keep taking chunks from the reader (it may be blocked handing one
                    errs.append(e)                   # over: leaving the loop here left it there for good — a run that ran out of HBM hung)
            done += s.n
            reads += s.n
    finally:
        q12.put(None)
        tr.join(); tw.join()
    if errs:
        raise errs[0]
    return time.perf_counter() - t0, reads
