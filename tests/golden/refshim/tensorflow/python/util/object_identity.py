"""`Reference`: hashable identity wrapper (tensorflow.python.util.object_identity), used by gpflow.utilities.traversal."""


class Reference:
    def __init__(self, wrapped):
        self._wrapped = wrapped

    def deref(self):
        return self._wrapped

    def __hash__(self):
        return id(self._wrapped)

    def __eq__(self, other):
        return isinstance(other, Reference) and other._wrapped is self._wrapped
