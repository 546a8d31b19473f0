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


# ---------------------------------------------------------------------------
# The reference's own convention (sailfish/kernel/parse_api.py:29-84, doc/source/kernels.rst:138-157): a kernel is
# ``PUBLIC void name(`` followed by its arguments, one per line, each ``<type> <name>`` closed by ``,`` or ``)`` with an
# optional ``// :: <constraint>`` comment.  Solvers hand that text to ``Library(code, ...)``; in the hip mode nothing is
# compiled from it, but it says what the caller is about to call -- ``Library`` checks it against the header.
# ---------------------------------------------------------------------------
_REF_START = re.compile(r"^\s*PUBLIC\s+void\s+(\w+)\s*\((.*)$")
_REF_ARG = re.compile(r"^\s*(?:const\s+)?(int|double)\s*(\*?)\s*(\w+)\s*$")


def parse_reference_api(code):
    """
    ``{name: Symbol}`` of the ``PUBLIC void`` kernels declared in reference-convention source text.  Constraints are
    kept as written (``$`` = the argument).  A declaration whose argument list cannot be read raises ValueError.
    """
    api = {}
    text = code.splitlines()
    k = 0
    while k < len(text):
        m = _REF_START.match(text[k])
        k += 1
        if not m:
            continue
        name, rest = m.group(1), m.group(2)
        args, done = [], False
        pending = [rest] if rest.strip() else []
        while not done:
            if not pending:
                if k >= len(text):
                    raise ValueError(f"declaration of {name} is not closed")
                pending.append(text[k])
                k += 1
            line = pending.pop(0)
            decl, _, comment = line.partition("//")
            constraint = comment.partition("::")[2].strip()
            for piece in filter(None, (p.strip() for p in re.split(r"([,)])", decl))):
                if piece == ",":
                    continue
                if piece == ")":
                    done = True
                    break
                a = _REF_ARG.match(piece)
                if not a:
                    raise ValueError(f"cannot parse argument {piece!r} of {name}")
                args.append(Argument(a.group(1) + a.group(2), a.group(3), constraint))
        api[name] = Symbol(name, args)
    return api


def signature_mismatch(given, ours):
    """None when the two Symbols take the same (type, name) arguments in the same order, else a description."""
    a = [(x.dtype, x.name) for x in given.args]
    b = [(x.dtype, x.name) for x in ours.args]
    if a == b:
        return None
    if len(a) != len(b):
        return f"{given.name}: the code declares {len(a)} arguments, the library takes {len(b)}"
    n = next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)
    return (f"{given.name}: argument {n} is `{a[n][0]} {a[n][1]}` in the code and `{b[n][0]} {b[n][1]}` in the library")
