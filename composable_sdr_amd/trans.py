"""Stream/array combinators of src/ComposableSDR/Trans.hs that feed and drain the DSP
blocks: takeNArr (:33-56), compact (:58-84), distribute_ (:106-117), mix (:119-122),
mux (:124-129).  Streams are Python iterables of numpy arrays; a Fold is an object with
step(a) / done() (streamly's Fold step/extract)."""
import numpy as np

from .pipes import Pipe


def takeNArr(n, stream):
    """Pass arrays until n samples in total went by, trimming the last one (Trans.hs:33-56)."""
    seen = 0
    for a in stream:
        togo = n - seen
        if togo == 0:
            return
        if togo >= len(a):
            seen += len(a)
            yield a
        else:
            seen = n
            yield a[:togo]


class Fold:
    def step(self, a):
        raise NotImplementedError

    def done(self):
        return None


class compact(Fold):
    """compact n fold (Trans.hs:58-84): buffer until >= n samples, emit EXACTLY n, keep the
    rest; at end of stream push the remainder downstream even when it is empty."""

    def __init__(self, n, downstream):
        self.n, self.down = n, downstream
        self.buf = None

    def step(self, a):
        b = a if self.buf is None or len(self.buf) == 0 else np.concatenate([self.buf, a])
        if len(b) >= self.n:
            self.down.step(b[:self.n])
            self.buf = b[self.n:]
        else:
            self.buf = b
        return self

    def done(self):
        rest = self.buf if self.buf is not None else np.empty(0, dtype=np.complex64)
        self.down.step(rest)
        return self.down.done()


class addPipe(Fold):
    """addPipe pipe fold (Types.hs:117-131): downstream start first, then create; on done
    destroy the pipe's resource, then finish downstream."""

    def __init__(self, pipe, downstream):
        self.down = downstream
        self.pipe = pipe
        self.r = pipe._start()

    def step(self, a):
        self.down.step(self.pipe._process(self.r, a))
        return self

    def done(self):
        self.pipe._done(self.r)
        return self.down.done()


class distribute_(Fold):
    """distribute_ folds (Trans.hs:106-117): element k of the input list goes to fold k
    (zip semantics: extra elements or extra folds are ignored)."""

    def __init__(self, folds):
        self.folds = list(folds)

    def step(self, arrays):
        for f, a in zip(self.folds, arrays):
            f.step(a)
        return self

    def done(self):
        for f in self.folds:
            f.done()


class collect(Fold):
    """Test/sink helper: keeps every array it is fed (stands in for fileSink)."""

    def __init__(self):
        self.items = []

    def step(self, a):
        self.items.append(np.array(a, copy=True))
        return self

    def done(self):
        return self.items

    def concat(self):
        return np.concatenate(self.items) if self.items else np.empty(0)


# mix (Trans.hs:119-122): foldl1 of element-wise (+) over the channel list
def _mix_process(_, arrays):
    acc = arrays[0]
    for a in arrays[1:]:
        n = min(len(acc), len(a))               # zipWith truncates to the shorter list
        acc = (acc[:n] + a[:n]).astype(acc.dtype)
    return acc


mix = Pipe(lambda: None, _mix_process, lambda r: None)


def mux(ps):
    """mux ps (Trans.hs:124-129): pipe k on list element k, one state per pipe."""
    def start():
        return [p._start() for p in ps]

    def process(rs, arrays):
        return [p._process(r, a) for p, r, a in zip(ps, rs, arrays)]

    def done(rs):
        for p, r in zip(ps, rs):
            p._done(r)
    return Pipe(start, process, done)
