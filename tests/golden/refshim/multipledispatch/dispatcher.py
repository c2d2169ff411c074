import itertools


def str_signature(sig):
    return ", ".join(getattr(c, "__name__", str(c)) for c in sig)


def variadic_signature_matches(types, full_signature):
    return False


def _supercedes(a, b):
    """signature a is at least as specific as b"""
    return len(a) == len(b) and all(issubclass(x, y) for x, y in zip(a, b))


def _ordering(signatures):
    """most specific first (topological order of the `supercedes` relation, ties in registration order)"""
    sigs = list(signatures)
    out = []
    remaining = list(sigs)
    while remaining:
        for s in remaining:
            # s can go next if no OTHER remaining signature is strictly more specific than s
            if not any(o is not s and _supercedes(o, s) and not _supercedes(s, o) for o in remaining):
                out.append(s)
                remaining.remove(s)
                break
        else:  # cycle (cannot happen with a partial order)
            out.extend(remaining)
            break
    return out


class Dispatcher:
    def __init__(self, name, doc=None):
        self.name = self.__name__ = name
        self.funcs = {}
        self.doc = doc
        self._cache = {}
        self._ordering = None

    def register(self, *types, **kwargs):
        def _(func):
            self.add(types, func)
            return func
        return _

    def add(self, signature, func):
        # a tuple inside a signature position means "any of these"
        if any(isinstance(t, tuple) for t in signature):
            for typs in itertools.product(*[t if isinstance(t, tuple) else (t,) for t in signature]):
                self.add(typs, func)
            return
        self.funcs[tuple(signature)] = func
        self._cache.clear()
        self._ordering = None

    @property
    def ordering(self):
        if self._ordering is None:
            self._ordering = _ordering(self.funcs)
        return self._ordering

    def dispatch(self, *types):
        if types in self.funcs:
            return self.funcs[types]
        for sig in self.ordering:
            if len(sig) == len(types) and all(map(issubclass, types, sig)):
                return self.funcs[sig]
        return None

    def __call__(self, *args, **kwargs):
        types = tuple(type(a) for a in args)
        func = self._cache.get(types)
        if func is None:
            func = self.dispatch(*types)
            if func is None:
                raise NotImplementedError(f"Could not find signature for {self.name}: <{str_signature(types)}>")
            self._cache[types] = func
        return func(*args, **kwargs)
