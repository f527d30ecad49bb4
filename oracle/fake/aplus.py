"""Test-infrastructure stand-in for the `aplus` Promises/A+ package (absent here).

Only the surface vaex's promise.py / delayed.py use. Needed solely so that the
golden-vector generator can `import vaex` from /root/reference in this container.
"""
import threading


def _isFunction(v):
    return callable(v)


def _isPromise(obj):
    return isinstance(obj, Promise) or (hasattr(obj, "done") and callable(getattr(obj, "done")))


class Promise:
    PENDING, REJECTED, FULFILLED = -1, 0, 1

    def __init__(self):
        self._state = self.PENDING
        self.value = None
        self.reason = None
        self._cb_lock = threading.RLock()
        self._callbacks = []
        self._errbacks = []

    @staticmethod
    def fulfilled(x):
        p = Promise()
        p.fulfill(x)
        return p

    @staticmethod
    def rejected(reason):
        p = Promise()
        p.reject(reason)
        return p

    def fulfill(self, x):
        if self is x:
            raise TypeError("Cannot resolve promise with itself.")
        elif _isPromise(x):
            x.done(self.fulfill, self.reject)
        else:
            self._fulfill(x)

    def _fulfill(self, value):
        with self._cb_lock:
            if self._state != self.PENDING:
                return
            self.value = value
            self._state = self.FULFILLED
            callbacks, self._callbacks = self._callbacks, None
        for cb in callbacks:
            try:
                cb(value)
            except Exception:
                pass

    def reject(self, reason):
        assert isinstance(reason, BaseException), reason
        with self._cb_lock:
            if self._state != self.PENDING:
                return
            self.reason = reason
            self._state = self.REJECTED
            errbacks, self._errbacks = self._errbacks, None
        for cb in errbacks:
            try:
                cb(reason)
            except Exception:
                pass

    @property
    def isPending(self):
        return self._state == self.PENDING

    @property
    def isFulfilled(self):
        return self._state == self.FULFILLED

    @property
    def isRejected(self):
        return self._state == self.REJECTED

    def get(self, timeout=None):
        if self._state == self.PENDING:
            raise ValueError("Value not available, promise is still pending")
        if self._state == self.FULFILLED:
            return self.value
        raise self.reason

    def done(self, success=None, failure=None):
        with self._cb_lock:
            if success is not None:
                self.addCallback(success)
            if failure is not None:
                self.addErrback(failure)

    def addCallback(self, f):
        with self._cb_lock:
            if self._state == self.PENDING:
                self._callbacks.append(f)
                return
        if self._state == self.FULFILLED:
            f(self.value)

    def addErrback(self, f):
        with self._cb_lock:
            if self._state == self.PENDING:
                self._errbacks.append(f)
                return
        if self._state == self.REJECTED:
            f(self.reason)

    def then(self, success=None, failure=None):
        ret = Promise()

        def ok(v):
            try:
                ret.fulfill(success(v) if _isFunction(success) else v)
            except Exception as e:
                ret.reject(e)

        def bad(r):
            try:
                if _isFunction(failure):
                    ret.fulfill(failure(r))
                else:
                    ret.reject(r)
            except Exception as e:
                ret.reject(e)

        self.done(ok, bad)
        return ret


def listPromise(*promises):
    import sys
    P = sys.modules[__name__].Promise
    ret = P()
    promises = list(promises)
    if not promises:
        ret.fulfill([])
        return ret

    def check(_=None):
        if all(p.isFulfilled for p in promises):
            ret.fulfill([p.value for p in promises])

    for p in promises:
        p.done(check, ret.reject)
    return ret


def dictPromise(m):
    keys = list(m.keys())
    return listPromise(*[m[k] for k in keys]).then(lambda vals: dict(zip(keys, vals)))
