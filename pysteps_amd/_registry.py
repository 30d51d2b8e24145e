"""Tiny case-insensitive name -> callable table used by the interface mirrors."""


class MethodTable:
    """Lookup with the semantics of pysteps' ``get_method`` helpers.

    Keys are lower-cased when they are strings; ``None`` is a legal key; an
    unknown key raises ``ValueError`` listing what is available (reference
    behaviour: motion/interface.py:97-111, extrapolation/interface.py:134-145).
    """

    def __init__(self, kind):
        self.kind = kind
        self._table = {}

    @staticmethod
    def _norm(name):
        return name.lower() if isinstance(name, str) else name

    def add(self, names, fn):
        for name in names if isinstance(names, (list, tuple)) else [names]:
            self._table[self._norm(name)] = fn
        return fn

    def names(self):
        return list(self._table.keys())

    def __contains__(self, name):
        return self._norm(name) in self._table

    def lookup(self, name):
        key = self._norm(name)
        try:
            return self._table[key]
        except (KeyError, TypeError):
            raise ValueError(
                "Unknown {} method {}\nThe available methods are:{}".format(
                    self.kind, name, self.names()
                )
            ) from None
