"""
Reads the C-ABI declarations of ``include/sfhip.h`` into a signature table.

Mirrors what the reference's ``sailfish/kernel/parse_api.py:8-84`` extracts from
kernel source text (``Symbol(name, [Argument(dtype, name, constraint)])``, rank =
number of leading ``int`` arguments, 1 <= rank <= 3), but the text parsed here is
our own header: every kernel entry point is declared as ``SFHIP_API int name(``
followed by one argument per line with an optional ``// :: <python constraint>``
comment, ``$`` standing for the argument itself.
"""
import re
from typing import List, NamedTuple


class Argument(NamedTuple):
    dtype: str  # "int" | "double" | "double*"
    name: str
    constraint: str


class Symbol(NamedTuple):
    name: str
    args: List[Argument]

    @property
    def rank(self):
        r = 0
        for arg in self.args:
            if arg.dtype != "int":
                break
            r += 1
        return r


_START = re.compile(r"^\s*SFHIP_API\s+int\s+(\w+)\s*\(\s*$")
_ARG = re.compile(r"^\s*(int|double)\s*(\*?)\s*(\w+)\s*([,)])\s*(?://(.*))?$")
_KNOWN = ("int", "double", "double*")


def parse_api(text):
    """
    Return ``{name: Symbol}`` for every ``SFHIP_API`` declaration in `text`.

    Raises ValueError for a declaration whose rank is not 1, 2 or 3 or whose
    argument line cannot be understood (the reference silently stops at such a
    line, ``parse_api.py:53-55``; a header we control should never contain one).
    """
    api = {}
    lines = iter(text.splitlines())
    for line in lines:
        m = _START.match(line)
        if not m:
            continue
        name, args = m.group(1), []
        for line in lines:
            a = _ARG.match(line)
            if not a:
                raise ValueError(f"cannot parse argument line of {name}: {line!r}")
            base, star, argname, closer, comment = a.groups()
            dtype = base + star
            if dtype not in _KNOWN:
                raise ValueError(f"unsupported type {dtype} in {name}")
            constraint = (comment or "").partition("::")[2].strip()
            args.append(Argument(dtype, argname, constraint))
            if closer == ")":
                break
        sym = Symbol(name, args)
        if not 1 <= sym.rank <= 3:
            raise ValueError(f"kernel {name} has rank {sym.rank}, must be 1, 2, or 3")
        api[name] = sym
    return api
