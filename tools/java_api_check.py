#!/usr/bin/env python3
"""Compile-substitute for integration/java (there is no JDK in this image, VERDICT r3 #3b).

Resolves what the plug-in's Java sources USE of the reference — imported classes, nested types, constructors, static and instance
methods along call chains, fields / enum constants, `case` labels of enum switches, @Override targets and the abstract methods a
concrete class must implement — against the reference's own SOURCES (a small Java declaration parser, no javac):

    name + arity for every call, parameter types wherever the argument's static type can be derived from declarations
    (locals, parameters, fields, casts, `new`, literals, return types of resolved calls).

What it cannot see is reported, not guessed: calls on types outside /root/reference (java.*, fastutil, clearspring, roaringbitmap) are
counted as `external`, receivers whose type cannot be derived as `untyped`.  Exit code 1 when a reference-resolvable use does not exist.

Usage: tools/java_api_check.py [--reference /root/reference] [--sources integration/java] [-v]
"""
import argparse
import os
import re
import sys

PRIMS = {"int", "long", "double", "float", "boolean", "byte", "short", "char", "void"}
BOX = {"int": "Integer", "long": "Long", "double": "Double", "float": "Float", "boolean": "Boolean", "byte": "Byte", "short": "Short",
       "char": "Character"}
UNBOX = {v: k for k, v in BOX.items()}
WIDEN = {"byte": {"short", "int", "long", "float", "double"}, "short": {"int", "long", "float", "double"},
         "char": {"int", "long", "float", "double"}, "int": {"long", "float", "double"}, "long": {"float", "double"},
         "float": {"double"}}
MODS = {"public", "protected", "private", "static", "final", "abstract", "native", "synchronized", "transient", "volatile", "default",
        "strictfp", "sealed", "non-sealed"}

TOKEN_RE = re.compile(r"""
    (?P<ws>\s+) | (?P<lc>//[^\n]*) | (?P<bc>/\*.*?\*/) |
    (?P<str>"(?:\\.|[^"\\])*") | (?P<chr>'(?:\\.|[^'\\])*') |
    (?P<num>(?:0[xX][0-9a-fA-F_]+|\d[\d_]*\.?\d*(?:[eE][+-]?\d+)?)[lLfFdD]?) |
    (?P<id>[A-Za-z_$][A-Za-z_$0-9]*) |
    (?P<op>->|::|\.\.\.|>>>=|<<=|>>=|\+\+|--|&&|\|\||==|!=|<=|>=|\+=|-=|\*=|/=|&=|\|=|\^=|%=|[{}()\[\];,.@=<>!~?:+\-*/&|^%])
""", re.X | re.S)


def tokenize(text):
    out = []
    for m in TOKEN_RE.finditer(text):
        k = m.lastgroup
        if k in ("ws", "lc", "bc"):
            continue
        out.append((k, m.group()))
    return out


class Member:
    def __init__(self, name, ret, params, mods, varargs=False, body=None):
        self.name, self.ret, self.params, self.mods, self.varargs, self.body = name, ret, params, mods, varargs, body


class Klass:
    def __init__(self, name, kind, outer=None):
        self.name, self.kind, self.outer = name, kind, outer
        self.extends, self.implements = [], []
        self.type_params = []
        self.methods, self.ctors, self.fields, self.nested, self.enum_constants = {}, [], {}, {}, []
        self.file = None   # JavaFile

    def fq(self):
        return (self.outer.fq() + "." if self.outer else (self.file.package + "." if self.file.package else "")) + self.name


class JavaFile:
    def __init__(self, path):
        self.path = path
        self.package = ""
        self.imports = {}      # simple name -> fq
        self.star_imports = []
        self.static_imports = {}
        self.classes = {}


def skip_balanced(toks, i, open_, close_):
    """toks[i] is open_; returns the index after its matching close_."""
    depth = 0
    while i < len(toks):
        t = toks[i][1]
        if t == open_:
            depth += 1
        elif t == close_:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    return i


def skip_generic(toks, i):
    """toks[i] is '<' of a type argument list; returns the index after the matching '>'."""
    depth = 0
    while i < len(toks):
        t = toks[i][1]
        if t == "<":
            depth += 1
        elif t == ">":
            depth -= 1
            if depth == 0:
                return i + 1
        elif t in (";", "{", "(", ")") :
            return i
        i += 1
    return i


def parse_type(toks, i):
    """Parses a type at toks[i]; returns (base name with dots, index after) or (None, i).  Generic arguments and array dims are dropped
    from the name but '[]' is appended."""
    while i < len(toks) and toks[i][1] == "@":   # type annotations
        i += 2
        if i < len(toks) and toks[i][1] == "(":
            i = skip_balanced(toks, i, "(", ")")
    if i >= len(toks) or toks[i][0] != "id":
        return None, i
    name = toks[i][1]
    i += 1
    while True:
        if i < len(toks) and toks[i][1] == "<":
            j = skip_generic(toks, i)
            if j == i:
                break
            i = j
        if i + 1 < len(toks) and toks[i][1] == "." and toks[i + 1][0] == "id":
            name += "." + toks[i + 1][1]
            i += 2
            continue
        break
    while i + 1 < len(toks) and toks[i][1] == "[" and toks[i + 1][1] == "]":
        name += "[]"
        i += 2
    if i < len(toks) and toks[i][1] == "...":
        name += "..."
        i += 1
    return name, i


def parse_params(toks, i):
    """toks[i] == '('; returns ([(type, name)], varargs, index after ')')."""
    end = skip_balanced(toks, i, "(", ")")
    params, varargs = [], False
    j = i + 1
    while j < end - 1:
        while toks[j][1] in ("final",) or toks[j][1] == "@":
            if toks[j][1] == "@":
                j += 2
                if toks[j][1] == "(":
                    j = skip_balanced(toks, j, "(", ")")
            else:
                j += 1
        ty, j2 = parse_type(toks, j)
        if ty is None:
            break
        if ty.endswith("..."):
            varargs = True
            ty = ty[:-3] + "[]"
        pname = toks[j2][1] if j2 < end - 1 and toks[j2][0] == "id" else ""
        j = j2 + 1
        while j < end - 1 and toks[j][1] == "[":   # int x[]
            ty += "[]"
            j += 2
        params.append((ty, pname))
        if j < end - 1 and toks[j][1] == ",":
            j += 1
    return params, varargs, end


def parse_class_body(toks, i, klass):
    """toks[i] == '{' of the class body; fills klass; returns index after '}'."""
    end = skip_balanced(toks, i, "{", "}")
    j = i + 1
    if klass.kind == "enum":   # constants up to ';' or the end
        while j < end - 1:
            if toks[j][1] == "@":
                j += 2
                if toks[j][1] == "(":
                    j = skip_balanced(toks, j, "(", ")")
                continue
            if toks[j][0] == "id":
                klass.enum_constants.append(toks[j][1])
                klass.fields[toks[j][1]] = klass.name
                j += 1
                if j < end - 1 and toks[j][1] == "(":
                    j = skip_balanced(toks, j, "(", ")")
                if j < end - 1 and toks[j][1] == "{":
                    j = skip_balanced(toks, j, "{", "}")
                if j < end - 1 and toks[j][1] == ",":
                    j += 1
                    continue
            if j < end - 1 and toks[j][1] == ";":
                j += 1
            break
    while j < end - 1:
        t = toks[j][1]
        if t == ";":
            j += 1
            continue
        mods = set()
        while j < end - 1:
            t = toks[j][1]
            if t == "@":
                if toks[j + 1][1] == "interface":
                    break
                mods.add("@" + toks[j + 1][1])
                j += 2
                while j < end - 1 and toks[j][1] == ".":   # @a.b.C
                    j += 2
                if j < end - 1 and toks[j][1] == "(":
                    j = skip_balanced(toks, j, "(", ")")
            elif t in MODS:
                mods.add(t)
                j += 1
            else:
                break
        if j >= end - 1:
            break
        t = toks[j][1]
        if t == "{":   # initializer block
            j = skip_balanced(toks, j, "{", "}")
            continue
        if t in ("class", "interface", "enum", "record") or (t == "@" and toks[j + 1][1] == "interface"):
            if t == "@":
                j += 1
                t = "interface"
            j = parse_class_decl(toks, j, klass.file, klass, t)
            continue
        tparams = []
        if t == "<":   # generic method
            k = skip_generic(toks, j)
            tparams = [x[1] for x in toks[j + 1:k - 1] if x[0] == "id"]
            j = k
        # constructor?
        if toks[j][0] == "id" and toks[j][1] == klass.name and toks[j + 1][1] == "(":
            params, varargs, k = parse_params(toks, j + 1)
            while k < end - 1 and toks[k][1] not in ("{", ";"):
                k += 1
            body = None
            if toks[k][1] == "{":
                k2 = skip_balanced(toks, k, "{", "}")
                body = (k, k2)
                k = k2
            else:
                k += 1
            klass.ctors.append(Member(klass.name, klass.name, params, mods, varargs, body))
            j = k
            continue
        ty, k = parse_type(toks, j)
        if ty is None or k >= end - 1 or toks[k][0] != "id":
            # not a member we understand: skip to the next ';' or balanced block
            while j < end - 1 and toks[j][1] not in (";", "{"):
                j += 1
            j = skip_balanced(toks, j, "{", "}") if j < end - 1 and toks[j][1] == "{" else j + 1
            continue
        name = toks[k][1]
        k += 1
        if toks[k][1] == "(":   # method
            params, varargs, k = parse_params(toks, k)
            while k < end - 1 and toks[k][1] == "[":
                ty += "[]"
                k += 2
            while k < end - 1 and toks[k][1] not in ("{", ";"):
                k += 1
            body = None
            if toks[k][1] == "{":
                k2 = skip_balanced(toks, k, "{", "}")
                body = (k, k2)
                k = k2
            else:
                k += 1
            if klass.kind == "interface" and body is None and "static" not in mods:
                mods.add("abstract")
            if ty in tparams or ty in klass.type_params:
                ty = "?"
            m = Member(name, ty, params, mods, varargs, body)
            m.tparams = tparams
            klass.methods.setdefault(name, []).append(m)
            j = k
            continue
        # field(s)
        while True:
            fty = ty
            while toks[k][1] == "[":
                fty += "[]"
                k += 2
            klass.fields[name] = "?" if fty in klass.type_params else fty
            if klass.kind == "interface":
                mods.add("static")
            if toks[k][1] == "=":
                depth = 0
                while k < end - 1:
                    tt = toks[k][1]
                    if tt in ("(", "{", "["):
                        depth += 1
                    elif tt in (")", "}", "]"):
                        depth -= 1
                    elif tt in (",", ";") and depth == 0:
                        break
                    k += 1
            if toks[k][1] == ",":
                name = toks[k + 1][1]
                k += 2
                continue
            break
        j = k + 1
    return end


def parse_class_decl(toks, i, jf, outer, kind):
    """toks[i] is class/interface/enum/record; returns index after the body."""
    name = toks[i + 1][1]
    k = Klass(name, "interface" if kind == "interface" else kind, outer)
    k.file = jf
    j = i + 2
    if toks[j][1] == "<":
        e = skip_generic(toks, j)
        depth = 0
        for x in range(j, e):
            if toks[x][1] == "<":
                depth += 1
            elif toks[x][1] == ">":
                depth -= 1
            elif depth == 1 and toks[x][0] == "id" and toks[x - 1][1] in ("<", ","):
                k.type_params.append(toks[x][1])
        j = e
    if kind == "record" and toks[j][1] == "(":
        params, _, j = parse_params(toks, j)
        k.ctors.append(Member(name, name, params, {"public"}))
        for ty, pn in params:
            k.methods.setdefault(pn, []).append(Member(pn, ty, [], {"public"}))
    while toks[j][1] != "{":
        if toks[j][1] == "extends":
            j += 1
            while True:
                ty, j = parse_type(toks, j)
                k.extends.append(ty)
                if toks[j][1] == ",":
                    j += 1
                    continue
                break
        elif toks[j][1] in ("implements", "permits"):
            which = toks[j][1]
            j += 1
            while True:
                ty, j = parse_type(toks, j)
                if which == "implements":
                    k.implements.append(ty)
                if toks[j][1] == ",":
                    j += 1
                    continue
                break
        else:
            j += 1
    if outer is not None:
        outer.nested[name] = k
    else:
        jf.classes[name] = k
    return parse_class_body(toks, j, k)


def parse_java(path):
    with open(path, encoding="utf-8", errors="replace") as f:
        text = f.read()
    toks = tokenize(text)
    jf = JavaFile(path)
    jf.toks = toks
    i = 0
    n = len(toks)
    while i < n:
        t = toks[i][1]
        if t == "package":
            j = i + 1
            parts = []
            while toks[j][1] != ";":
                parts.append(toks[j][1])
                j += 1
            jf.package = "".join(parts)
            i = j + 1
        elif t == "import":
            j = i + 1
            static = toks[j][1] == "static"
            if static:
                j += 1
            parts = []
            while toks[j][1] != ";":
                parts.append(toks[j][1])
                j += 1
            fq = "".join(parts)
            if fq.endswith(".*"):
                (jf.star_imports if not static else jf.star_imports).append(fq[:-2])
            elif static:
                jf.static_imports[fq.rsplit(".", 1)[1]] = fq.rsplit(".", 1)[0]
            else:
                jf.imports[fq.rsplit(".", 1)[1]] = fq
            i = j + 1
        elif t in ("class", "interface", "enum", "record") and (i + 1 < n and toks[i + 1][0] == "id"):
            i = parse_class_decl(toks, i, jf, None, t)
        elif t == "@" and i + 1 < n and toks[i + 1][1] == "interface":
            i = parse_class_decl(toks, i + 1, jf, None, "interface")
        else:
            i += 1
    return jf


class World:
    """Index of the reference's sources (+ the plug-in's own) by fully qualified name."""

    def __init__(self, reference, own_root):
        self.by_fq_path = {}
        self.files = {}
        self._cache, self._busy = {}, set()
        for root in (reference, own_root):
            for dp, dn, fn in os.walk(root):
                if "/src/test/" in dp + "/" and root == reference:
                    continue
                for f in fn:
                    if not f.endswith(".java"):
                        continue
                    p = os.path.join(dp, f)
                    m = re.search(r"/(?:java|gen-java|generated-sources/[^/]+)/(.+)\.java$", p) if root == reference else None
                    if m:
                        fq = m.group(1).replace("/", ".")
                    elif root == own_root:
                        fq = os.path.relpath(p, own_root)[:-5].replace("/", ".")
                    else:
                        continue
                    self.by_fq_path.setdefault(fq, p)

    def file_of(self, fq):
        p = self.by_fq_path.get(fq)
        if not p:
            return None
        if p not in self.files:
            self.files[p] = parse_java(p)
        return self.files[p]

    def klass(self, fq):
        """fq may name a nested class: a.b.Outer.Inner"""
        parts = fq.split(".")
        for cut in range(len(parts), 0, -1):
            jf = self.file_of(".".join(parts[:cut]))
            if jf is None:
                continue
            k = jf.classes.get(parts[cut - 1])
            for nm in parts[cut:]:
                k = self.find_nested(k, nm) if k else None
            return k
        return None

    def find_nested(self, k, name, seen=None):
        seen = seen or set()
        if k is None or id(k) in seen:
            return None
        seen.add(id(k))
        if name in k.nested:
            return k.nested[name]
        for sup in k.extends + k.implements:
            sk = self.resolve(sup, k)
            r = self.find_nested(sk, name, seen) if sk else None
            if r:
                return r
        return None

    def resolve(self, tname, ctx):
        """Type name as written inside class ctx -> Klass, or None (external / unknown)."""
        if tname is None:
            return None
        key = (id(ctx), tname)
        if key in self._cache:
            return self._cache[key]
        if key in self._busy:   # a supertype's name being resolved through the inherited nested types of the same class
            return None
        self._busy.add(key)
        try:
            r = self._resolve(tname, ctx)
        finally:
            self._busy.discard(key)
        if not self._busy:
            self._cache[key] = r
        return r

    def _resolve(self, tname, ctx):
        tname = tname.replace("[]", "")
        if tname in PRIMS or tname == "?":
            return None
        first, _, rest = tname.partition(".")
        # enclosing classes and their nested / inherited nested types
        k = ctx
        while k is not None:
            if k.name == first:
                cand = k
            else:
                cand = self.find_nested(k, first)
            if cand:
                for nm in rest.split(".") if rest else []:
                    cand = self.find_nested(cand, nm)
                    if cand is None:
                        break
                if cand:
                    return cand
            k = k.outer
        jf = ctx.file
        if first in jf.classes:
            cand = jf.classes[first]
        elif first in jf.imports:
            cand = self.klass(jf.imports[first])
        else:
            cand = self.klass((jf.package + "." if jf.package else "") + first)
            if cand is None:
                for sp in jf.star_imports:
                    cand = self.klass(sp + "." + first)
                    if cand:
                        break
            if cand is None and rest:   # fully qualified
                return self.klass(tname)
        if cand is None:
            return None
        for nm in rest.split(".") if rest else []:
            cand = self.find_nested(cand, nm)
            if cand is None:
                return None
        return cand

    def supertypes(self, k, seen=None):
        seen = seen if seen is not None else {}
        for sup in k.extends + k.implements:
            sk = self.resolve(sup, k)
            if sk and id(sk) not in seen:
                seen[id(sk)] = sk
                self.supertypes(sk, seen)
        return list(seen.values())

    def external_super(self, k):
        """True when some supertype cannot be found in the sources (its members are then unknown)."""
        for c in [k] + self.supertypes(k):
            for sup in c.extends + c.implements:
                if self.resolve(sup, c) is None and sup.split("<")[0] not in ("Object",):
                    return True
        return False

    def methods(self, k, name):
        out = list(k.methods.get(name, []))
        for sk in self.supertypes(k):
            out += [(m) for m in sk.methods.get(name, [])]
        return out

    def field(self, k, name):
        for c in [k] + self.supertypes(k):
            if name in c.fields:
                return c.fields[name], c
        return None, None

    def is_subtype(self, k, target_name, ctx):
        if k is None:
            return None
        tk = self.resolve(target_name, ctx)
        if tk is None:
            return None
        return tk is k or any(s is tk for s in self.supertypes(k))


class Checker:
    def __init__(self, world, verbose=False):
        self.w = world
        self.verbose = verbose
        self.errors, self.checked, self.external, self.untyped = [], 0, 0, 0
        self.details = []

    def err(self, jf, msg):
        self.errors.append("%s: %s" % (os.path.relpath(jf.path), msg))

    def ok(self, what):
        self.checked += 1
        if self.verbose:
            self.details.append(what)

    # ---- types of expressions -----------------------------------------------------------------------------------------------
    def compatible(self, arg, param, ctx_param, ctx_arg):
        """arg / param: type names (or None = unknown).  True unless provably incompatible."""
        if arg is None or param is None or arg == "?" or param == "?":
            return True
        a, p = arg, param
        if a == "null":
            return p.replace("[]", "") not in PRIMS or p.endswith("[]")
        if a.count("[]") != p.count("[]"):
            return p in ("Object",) and True
        ab, pb = a.replace("[]", ""), p.replace("[]", "")
        ab, pb = ab.split(".")[-1] if False else ab, pb
        if ab == pb or ab.split(".")[-1] == pb.split(".")[-1]:
            return True
        if pb in ("Object", "T", "E", "K", "V", "R") or len(pb) == 1:
            return True
        if ab in PRIMS and pb in PRIMS:
            return pb in WIDEN.get(ab, set())
        if ab in PRIMS:
            return BOX[ab] == pb or pb in ("Number", "Comparable", "Serializable")
        if pb in PRIMS:
            return UNBOX.get(ab) == pb or pb in WIDEN.get(UNBOX.get(ab, ""), set())
        ak = self.w.resolve(ab, ctx_arg) if ctx_arg else None
        pk = self.w.resolve(pb, ctx_param) if ctx_param else None
        if ak is None or pk is None:
            return True
        if ak is pk or any(s is pk for s in self.w.supertypes(ak)):
            return True
        if self.w.external_super(ak):
            return True
        return False

    def pick(self, cands, args, ctx_arg):
        """cands: [(Member, declaring Klass)]; args: list of type names / None.  Returns (member, klass, reason)."""
        n = len(args)
        ar = [(m, k) for m, k in cands if len(m.params) == n or (m.varargs and n >= len(m.params) - 1)]
        if not ar:
            return None, None, "arity"
        for m, k in ar:
            good = True
            for idx, a in enumerate(args):
                pi = min(idx, len(m.params) - 1)
                pt = m.params[pi][0]
                if m.varargs and idx >= len(m.params) - 1:
                    # either the array itself or an element
                    if not (self.compatible(a, pt, k, ctx_arg) or self.compatible(a, pt[:-2], k, ctx_arg)):
                        good = False
                        break
                    continue
                if pt in getattr(m, "tparams", []) or pt in k.type_params:
                    continue
                if not self.compatible(a, pt, k, ctx_arg):
                    good = False
                    break
            if good:
                return m, k, None
        return None, None, "types"


class Scope:
    def __init__(self, parent=None):
        self.vars, self.parent = {}, parent

    def get(self, n):
        s = self
        while s:
            if n in s.vars:
                return s.vars[n]
            s = s.parent
        return None


class BodyWalker:
    """Walks one method body of a plug-in class: declarations -> scope, expressions -> checks."""

    def __init__(self, chk, jf, klass, toks):
        self.c, self.jf, self.k, self.t = chk, jf, klass, toks
        self.w = chk.w

    def where(self, i):
        return ""

    def type_exists(self, name):
        base = name.replace("[]", "")
        if base in PRIMS or base in ("var",):
            return True
        return None

    def walk_block(self, i, end, scope):
        """Statements in toks[i:end] (inside braces)."""
        t = self.t
        while i < end:
            x = t[i][1]
            if x == "{":
                e = skip_balanced(t, i, "{", "}")
                self.walk_block(i + 1, e - 1, Scope(scope))
                i = e
                continue
            if x in ("for",) and t[i + 1][1] == "(":
                e = skip_balanced(t, i + 1, "(", ")")
                inner = Scope(scope)
                # for (T v : expr) / for (T v = ...; ...; ...)
                j = i + 2
                if t[j][1] == "final":
                    j += 1
                ty, j2 = parse_type(t, j)
                if ty and j2 < e and t[j2][0] == "id" and t[j2 + 1][1] in (":", "=", ","):
                    inner.vars[t[j2][1]] = ty
                    self.note_type(ty, j)
                    k = j2 + 1
                    while k < e - 1 and t[k][1] == ",":   # int a = 0, b = 1
                        k += 1
                    self.walk_expr_range(j2 + 1, e - 1, inner)
                    # further declarators: "for (int i = 0, n = x; ...)"
                    depth = 0
                    for q in range(j2 + 1, e - 1):
                        if t[q][1] in ("(", "[", "{"):
                            depth += 1
                        elif t[q][1] in (")", "]", "}"):
                            depth -= 1
                        elif t[q][1] == ";" and depth == 0:
                            break
                        elif t[q][1] == "," and depth == 0 and t[q + 1][0] == "id" and t[q + 2][1] == "=":
                            inner.vars[t[q + 1][1]] = ty
                else:
                    self.walk_expr_range(i + 2, e - 1, inner)
                i = e
                if t[i][1] == "{":
                    e2 = skip_balanced(t, i, "{", "}")
                    self.walk_block(i + 1, e2 - 1, Scope(inner))
                    i = e2
                else:
                    e2 = self.stmt_end(i, end)
                    self.walk_block(i, e2, Scope(inner))
                    i = e2
                continue
            if x == "try" and t[i + 1][1] == "(":
                e = skip_balanced(t, i + 1, "(", ")")
                inner = Scope(scope)
                self.walk_block(i + 2, e - 1, inner)
                i = e
                if t[i][1] == "{":
                    e2 = skip_balanced(t, i, "{", "}")
                    self.walk_block(i + 1, e2 - 1, Scope(inner))
                    i = e2
                continue
            if x == "catch" and t[i + 1][1] == "(":
                e = skip_balanced(t, i + 1, "(", ")")
                inner = Scope(scope)
                j = i + 2
                if t[j][1] == "final":
                    j += 1
                ty, j2 = parse_type(t, j)
                while t[j2][1] == "|":
                    self.note_type(ty, j)
                    ty, j2 = parse_type(t, j2 + 1)
                if ty and t[j2][0] == "id":
                    inner.vars[t[j2][1]] = ty
                    self.note_type(ty, j)
                i = e
                if t[i][1] == "{":
                    e2 = skip_balanced(t, i, "{", "}")
                    self.walk_block(i + 1, e2 - 1, inner)
                    i = e2
                continue
            if x == "switch" and t[i + 1][1] == "(":
                e = skip_balanced(t, i + 1, "(", ")")
                sty = self.expr_type(i + 2, e - 1, scope)
                i = e
                if t[i][1] == "{":
                    e2 = skip_balanced(t, i, "{", "}")
                    self.walk_switch(i + 1, e2 - 1, Scope(scope), sty)
                    i = e2
                continue
            if x in ("if", "while", "synchronized") and t[i + 1][1] == "(":
                e = skip_balanced(t, i + 1, "(", ")")
                self.walk_expr_range(i + 2, e - 1, scope)
                i = e
                continue
            if x in ("else", "do", "try", "finally"):
                i += 1
                continue
            if x in ("return", "throw"):
                e = self.stmt_end(i, end)
                self.walk_expr_range(i + 1, e - 1, scope)
                i = e
                continue
            if x in ("break", "continue"):
                i = self.stmt_end(i, end)
                continue
            if x == ";":
                i += 1
                continue
            # local declaration?  [final] Type name (= ... | ; | ,)
            j = i
            if t[j][1] == "final":
                j += 1
            if t[j][0] == "id":
                ty, j2 = parse_type(t, j)
                if ty and j2 < end and t[j2][0] == "id" and t[j2 + 1][1] in ("=", ";", ",", "["):
                    e = self.stmt_end(i, end)
                    self.note_type(ty, j)
                    k = j2
                    while k < e:
                        name = t[k][1]
                        vty = ty
                        k += 1
                        while t[k][1] == "[":
                            vty += "[]"
                            k += 2
                        scope.vars[name] = vty
                        if t[k][1] == "=":
                            # initializer up to the top-level ',' or ';'
                            depth, q = 0, k + 1
                            while q < e:
                                if t[q][1] in ("(", "[", "{"):
                                    depth += 1
                                elif t[q][1] in (")", "]", "}"):
                                    depth -= 1
                                elif t[q][1] in (",", ";") and depth == 0:
                                    break
                                q += 1
                            ity = self.expr_type(k + 1, q, scope)
                            if ity is not None and not self.c.compatible(ity, vty, self.k, self.k):
                                self.c.err(self.jf, "initializer of `%s %s` has type %s" % (vty, name, ity))
                            k = q
                        if t[k][1] == ",":
                            k += 1
                            continue
                        break
                    i = e
                    continue
            e = self.stmt_end(i, end)
            self.walk_expr_range(i, e - 1 if t[e - 1][1] == ";" else e, scope)
            i = e

    def walk_switch(self, i, end, scope, sty):
        t = self.t
        sk = self.w.resolve(sty, self.k) if sty else None
        while i < end:
            if t[i][1] == "case":
                j = i + 1
                while t[j][1] not in (":", "->"):
                    j += 1
                if sk is not None and sk.kind == "enum":
                    for q in range(i + 1, j):
                        if t[q][0] == "id":
                            if t[q][1] in sk.enum_constants:
                                self.c.ok("enum constant %s.%s" % (sk.name, t[q][1]))
                            else:
                                self.c.err(self.jf, "case %s: not a constant of enum %s" % (t[q][1], sk.fq()))
                elif sk is None and sty is not None and sty not in PRIMS and sty != "String":
                    self.c.external += 1
                else:
                    self.walk_expr_range(i + 1, j, scope)
                i = j + 1
                continue
            if t[i][1] == "default":
                i += 2
                continue
            # statements until next case/default at depth 0
            j, depth = i, 0
            while j < end:
                if t[j][1] in ("{", "(", "["):
                    depth += 1
                elif t[j][1] in ("}", ")", "]"):
                    depth -= 1
                elif depth == 0 and t[j][1] in ("case", "default") and t[j - 1][1] in (";", "}", ":", "{"):
                    break
                j += 1
            self.walk_block(i, j, scope)
            i = j

    def stmt_end(self, i, end):
        """index after the ';' that ends the statement starting at i (balanced), or after a trailing block."""
        t = self.t
        depth = 0
        while i < end:
            x = t[i][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif x == ";" and depth == 0:
                return i + 1
            i += 1
        return end

    def note_type(self, ty, at):
        base = ty.replace("[]", "").replace("...", "")
        if base in PRIMS or base == "var":
            return
        k = self.w.resolve(base, self.k)
        if k is not None:
            self.c.ok("type %s" % k.fq())
            return
        first = base.split(".")[0]
        fq = self.jf.imports.get(first)
        if fq and fq.startswith("org.apache.pinot.") and not fq.startswith("org.apache.pinot.gpu."):
            self.c.err(self.jf, "type %s (import %s) not found in the reference" % (base, fq))
        elif base.startswith("org.apache.pinot."):
            self.c.err(self.jf, "type %s not found in the reference" % base)
        else:
            self.c.external += 1

    def walk_expr_range(self, i, end, scope):
        """Checks every call chain in toks[i:end] (an expression or an expression list)."""
        if i >= end:
            return
        # split at top-level commas / semicolons, type each piece (typing performs the checks)
        t = self.t
        depth, start = 0, i
        q = i
        while q < end:
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif x in (",", ";") and depth == 0:
                self.expr_type(start, q, scope)
                start = q + 1
            q += 1
        self.expr_type(start, end, scope)

    # ---- expression typing (performs the checks on the way) -------------------------------------------------------------------
    def split_top(self, i, end, seps):
        t = self.t
        depth, out, start = 0, [], i
        angle = 0
        for q in range(i, end):
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif depth == 0 and x in seps:
                out.append((start, q, x))
                start = q + 1
        out.append((start, end, None))
        return out

    def expr_type(self, i, end, scope):
        t = self.t
        if i >= end:
            return None
        # lambda:  x -> ..., (a, b) -> ..., () -> ...
        depth = 0
        for q in range(i, end):
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif x == "->" and depth == 0:
                inner = Scope(scope)
                for z in range(i, q):
                    if t[z][0] == "id":
                        inner.vars[t[z][1]] = None
                if t[q + 1][1] == "{":
                    self.walk_block(q + 2, end - 1, inner)
                else:
                    self.expr_type(q + 1, end, inner)
                return None
        # assignment (lowest precedence): type is the right side's; check both
        parts = self.split_top(i, end, {"=", "+=", "-=", "*=", "/=", "|=", "&=", "^=", "%=", "<<=", ">>=", ">>>="})
        if len(parts) > 1:
            self.expr_type(parts[0][0], parts[0][1], scope)
            ty = None
            for a, b, _ in parts[1:]:
                ty = self.expr_type(a, b, scope)
            return ty
        # ternary
        depth, qm = 0, -1
        for q in range(i, end):
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif x == "?" and depth == 0:
                qm = q
                break
        if qm >= 0:
            depth, colon = 0, -1
            nest = 0
            for q in range(qm + 1, end):
                x = t[q][1]
                if x in ("(", "[", "{"):
                    depth += 1
                elif x in (")", "]", "}"):
                    depth -= 1
                elif depth == 0 and x == "?":
                    nest += 1
                elif depth == 0 and x == ":":
                    if nest == 0:
                        colon = q
                        break
                    nest -= 1
            self.expr_type(i, qm, scope)
            if colon >= 0:
                a = self.expr_type(qm + 1, colon, scope)
                b = self.expr_type(colon + 1, end, scope)
                if a == b:
                    return a
                if a in PRIMS and b in PRIMS:
                    return a if b in WIDEN.get(a, set()) is False else (b if b in WIDEN.get(a, set()) else a)
                return a if b is None or b == "null" else (b if a is None or a == "null" else None)
            return None
        # binary operators: type each operand; result by a coarse rule
        for ops, res in (({"||", "&&"}, "boolean"), ({"==", "!=", "<=", ">=", "instanceof"}, "boolean"),):
            parts = self.split_top(i, end, ops)
            if len(parts) > 1:
                for a, b, sep in parts:
                    if a > i and t[a - 1][1] == "instanceof":
                        ty, _ = parse_type(t, a)
                        if ty:
                            self.note_type(ty, a)
                        continue
                    self.expr_type(a, b, scope)
                return res
        # relational < > (careful with generics: only when it does not parse as a generic type use) — treat as boolean if found at depth 0
        # between two operand-looking tokens
        depth = 0
        for q in range(i + 1, end - 1):
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif depth == 0 and x in ("<", ">") and (t[q + 1][1] == x or t[q - 1][1] == x) and not self.looks_generic(q, i, end):
                # a shift (the tokenizer leaves << >> >>> as single characters because of nested generics): int / long of the left side
                first = q if t[q - 1][1] != x else q - 1
                last = first
                while t[last + 1][1] == x:
                    last += 1
                lt = self.expr_type(i, first, scope)
                self.expr_type(last + 1, end, scope)
                return "long" if UNBOX.get(lt, lt) == "long" else "int"
            elif depth == 0 and x in ("<", ">") and not self.looks_generic(q, i, end):
                self.expr_type(i, q, scope)
                self.expr_type(q + 1, end, scope)
                return "boolean"
        for ops in ({"|"}, {"^"}, {"&"}, {"<<", ">>", ">>>"}, {"+", "-"}, {"*", "/", "%"}):
            parts = self.split_top_binary(i, end, ops)
            if len(parts) > 1:
                tys = [self.expr_type(a, b, scope) for a, b in parts]
                if "String" in tys:
                    return "String"
                if any(x is None for x in tys):
                    return None
                tys = [UNBOX.get(x, x) for x in tys]
                for cand in ("double", "float", "long"):
                    if cand in tys:
                        return cand
                return "int" if all(x in PRIMS for x in tys) else None
        return self.unary_type(i, end, scope)

    def looks_generic(self, q, i, end):
        t = self.t
        if t[q][1] == "<":
            e = skip_generic(t, q)
            if e <= end and e > q + 1 and t[e - 1][1] == ">":
                inside = t[q + 1:e - 1]
                if all(x[0] == "id" or x[1] in (",", ".", "?", "<", ">", "[", "]", "extends", "super") for x in inside):
                    return True
            return False
        # '>' : generic if a matching '<' opened earlier at this level
        depth = 0
        for z in range(q - 1, i - 1, -1):
            if t[z][1] == ">":
                depth += 1
            elif t[z][1] == "<":
                if depth == 0:
                    return self.looks_generic(z, i, end)
                depth -= 1
        return False

    def split_top_binary(self, i, end, ops):
        """split at binary occurrences of ops (not unary +/-)."""
        t = self.t
        depth, out, start = 0, [], i
        for q in range(i, end):
            x = t[q][1]
            if x in ("(", "[", "{"):
                depth += 1
            elif x in (")", "]", "}"):
                depth -= 1
            elif depth == 0 and x in ops and q > start:
                prev = t[q - 1]
                if prev[0] in ("id", "num", "str", "chr") or prev[1] in (")", "]"):
                    if x in ("<", ">") :
                        continue
                    out.append((start, q))
                    start = q + 1
        out.append((start, end))
        return out

    def unary_type(self, i, end, scope):
        t = self.t
        x = t[i][1]
        if x in ("!",):
            self.unary_type(i + 1, end, scope)
            return "boolean"
        if x in ("-", "+", "~", "++", "--"):
            return self.unary_type(i + 1, end, scope)
        if t[end - 1][1] in ("++", "--"):
            return self.unary_type(i, end - 1, scope)
        # cast: ( Type ) operand
        if x == "(":
            e = skip_balanced(t, i, "(", ")")
            if e < end:
                ty, j = parse_type(t, i + 1)
                if ty and j == e - 1 and (t[e][0] in ("id", "num", "str", "chr") or t[e][1] in ("(", "!", "~", "-")):
                    if not (t[e][1] == "-" and ty not in PRIMS):
                        self.note_type(ty, i + 1)
                        self.unary_type(e, end, scope)
                        return ty
        return self.chain_type(i, end, scope)

    def args_of(self, i, scope):
        """toks[i] == '('; returns ([arg types], index after ')')."""
        t = self.t
        e = skip_balanced(t, i, "(", ")")
        if e - 1 == i + 1:
            return [], e
        return [self.expr_type(a, b, scope) for a, b, _ in self.split_top(i + 1, e - 1, {","})], e

    def chain_type(self, i, end, scope):
        """primary followed by .name / .name(args) / [index] selectors."""
        t = self.t
        kind, x = t[i]
        cur = None      # current static type name (str) or None
        cur_k = None    # Klass of cur when it is in the sources
        static_ref = False
        ctx = self.k
        j = i
        if kind == "num":
            s = x.lower()
            cur = "long" if s.endswith("l") and not s.startswith("0x") or (s.startswith("0x") and s.endswith("l")) else (
                "float" if s.endswith("f") and not s.startswith("0x") else ("double" if ("." in s or ("e" in s and not s.startswith("0x")) or s.endswith("d") and not s.startswith("0x")) else "int"))
            j = i + 1
        elif kind == "str":
            cur = "String"
            j = i + 1
        elif kind == "chr":
            cur = "char"
            j = i + 1
        elif x in ("true", "false"):
            cur = "boolean"
            j = i + 1
        elif x == "null":
            cur = "null"
            j = i + 1
        elif x == "(":
            e = skip_balanced(t, i, "(", ")")
            cur = self.expr_type(i + 1, e - 1, scope)
            j = e
        elif x == "new":
            ty, j = parse_type(t, i + 1)
            if ty is None:
                return None
            # array creation: new T[n] / new T[] {...}
            if t[j][1] == "[" or ty.endswith("[]"):
                dims = ty.count("[]")
                base = ty.replace("[]", "")
                while j < end and t[j][1] == "[":
                    e = skip_balanced(t, j, "[", "]")
                    self.expr_type(j + 1, e - 1, scope)
                    dims += 1
                    j = e
                if j < end and t[j][1] == "{":
                    e = skip_balanced(t, j, "{", "}")
                    self.walk_expr_range(j + 1, e - 1, scope)
                    j = e
                self.note_type(base, i + 1)
                cur = base + "[]" * dims
            else:
                self.note_type(ty, i + 1)
                k = self.w.resolve(ty, self.k)
                args, j = self.args_of(j, scope)
                if k is not None:
                    if k.kind == "interface" or "abstract" in getattr(k, "mods", set()):
                        pass
                    cands = [(m, k) for m in k.ctors]
                    if not cands and not args:
                        self.c.ok("new %s() (implicit)" % k.name)
                    elif j < end and t[j][1] == "{" and k.kind == "interface":
                        self.c.ok("anonymous %s" % k.name)
                    else:
                        m, mk, why = self.c.pick(cands, args, self.k)
                        if m is None:
                            self.c.err(self.jf, "new %s(%s): no constructor matches (%s); declared: %s" % (
                                k.fq(), ", ".join(str(a) for a in args), why, "; ".join("(" + ", ".join(p[0] for p in c.params) + ")" for c in k.ctors) or "none"))
                        else:
                            self.c.ok("new %s/%d" % (k.fq(), len(args)))
                else:
                    self.c.external += 1
                if j < end and t[j][1] == "{":   # anonymous class body
                    e = skip_balanced(t, j, "{", "}")
                    anon = Klass("$anon", "class", self.k)
                    anon.file = self.jf
                    anon.extends = [ty]
                    parse_class_body(t, j, anon)
                    check_class(self.c, self.jf, anon, scope)
                    j = e
                cur = ty
        elif kind == "id":
            if x == "this":
                cur, j = self.k.name, i + 1
                cur_k = self.k
                if cur == "$anon":
                    cur = None
            elif x == "super":
                j = i + 1
                sk = self.w.resolve(self.k.extends[0], self.k) if self.k.extends else None
                cur_k, cur = sk, (self.k.extends[0] if self.k.extends else None)
                if j < end and t[j][1] == "(":   # explicit constructor invocation: super(args)
                    args, j = self.args_of(j, scope)
                    if sk is not None:
                        m, mk, why = self.c.pick([(c, sk) for c in sk.ctors], args, self.k)
                        if m is None and (sk.ctors or args):
                            self.c.err(self.jf, "super(%s): no constructor matches (%s); declared: %s" % (
                                ", ".join(str(a) for a in args), why, "; ".join("(" + ", ".join(p[0] for p in c.params) + ")" for c in sk.ctors) or "none"))
                        else:
                            self.c.ok("super/%d of %s" % (len(args), sk.fq()))
                    else:
                        self.c.external += 1
                    cur = None
            else:
                v = scope.get(x) if scope else None
                if (scope and self.in_scope(scope, x)) and not (i + 1 < end and t[i + 1][1] == "("):
                    cur, j = v, i + 1
                elif i + 1 < end and t[i + 1][1] == "(":
                    # unqualified method call: this class, its supertypes, enclosing classes
                    args, j = self.args_of(i + 1, scope)
                    cur = self.call_on(self.k, x, args, implicit=True)
                else:
                    fty, fk = self.find_field_enclosing(x)
                    if fty is not None:
                        cur, j = fty, i + 1
                        ctx = fk
                    else:
                        # a type name (possibly qualified: a.b.C or Outer.Inner)
                        ty, j2 = parse_type(t, i)
                        k, used = self.longest_type(i, end)
                        if k is not None:
                            cur_k, cur, j, static_ref = k, k.name, used, True
                            self.c.ok("type %s" % k.fq())
                        else:
                            # external type or unknown identifier
                            fq = self.jf.imports.get(x)
                            if fq and fq.startswith("org.apache.pinot.") and not fq.startswith("org.apache.pinot.gpu."):
                                self.c.err(self.jf, "type %s (import %s) not found in the reference" % (x, fq))
                            elif x == "org" and i + 4 < end and t[i + 2][1] == "apache" and t[i + 4][1] == "pinot":
                                self.c.err(self.jf, "qualified name starting at `%s` not found in the reference" % "".join(z[1] for z in t[i:min(end, i + 16)]))
                            self.c.external += 1
                            # swallow the rest of the chain
                            return None
        else:
            return None
        if cur_k is None and cur is not None:
            cur_k = self.w.resolve(cur.replace("[]", ""), ctx) if not cur.endswith("[]") else None
        # selectors
        while j < end:
            x = t[j][1]
            if x == "[":
                e = skip_balanced(t, j, "[", "]")
                self.expr_type(j + 1, e - 1, scope)
                if cur and cur.endswith("[]"):
                    cur = cur[:-2]
                    cur_k = self.w.resolve(cur, ctx) if not cur.endswith("[]") else None
                else:
                    cur, cur_k = None, None
                j = e
                static_ref = False
                continue
            if x == "." and j + 1 < end and t[j + 1][0] == "id" or (x == "." and j + 1 < end and t[j + 1][1] == "<"):
                if t[j + 1][1] == "<":   # explicit type arguments: a.<T>m()
                    j = skip_generic(t, j + 1) - 1
                name = t[j + 1][1]
                if j + 2 < end and t[j + 2][1] == "(":
                    args, e = self.args_of(j + 2, scope)
                    if cur_k is not None:
                        cur = self.call_on(cur_k, name, args, static_only=static_ref)
                        ctx = self.last_decl or cur_k
                    elif cur is not None and cur.endswith("[]"):
                        cur = None
                    else:
                        if cur is None:
                            self.c.untyped += 1
                        else:
                            self.c.external += 1
                            cur = self.external_call(cur, name, args)
                        if cur is None:
                            pass
                    cur_k = self.w.resolve(cur.replace("[]", ""), ctx) if cur and not cur.endswith("[]") else None
                    j = e
                    static_ref = False
                    continue
                # field / nested type / .length / .class
                if name == "length" and cur and cur.endswith("[]"):
                    cur, cur_k = "int", None
                elif name == "class":
                    cur, cur_k = "Class", None
                elif cur_k is not None:
                    fty, fk = self.w.field(cur_k, name)
                    if fty is not None:
                        self.c.ok("field %s.%s" % (cur_k.name, name))
                        cur = fty
                        ctx = fk
                        cur_k = self.w.resolve(cur.replace("[]", ""), fk) if not cur.endswith("[]") else None
                        static_ref = False
                    else:
                        nk = self.w.find_nested(cur_k, name)
                        if nk is not None:
                            cur_k, cur = nk, nk.name
                        elif self.w.external_super(cur_k):
                            self.c.external += 1
                            cur, cur_k = None, None
                        else:
                            self.c.err(self.jf, "%s has no field / nested type `%s`" % (cur_k.fq(), name))
                            cur, cur_k = None, None
                else:
                    if cur is None:
                        self.c.untyped += 1
                    else:
                        self.c.external += 1
                    cur = None
                j += 2
                continue
            if x == "::":
                return None
            break
        return cur

    last_decl = None

    def in_scope(self, scope, n):
        s = scope
        while s:
            if n in s.vars:
                return True
            s = s.parent
        return False

    def find_field_enclosing(self, name):
        k = self.k
        while k is not None:
            fty, fk = self.w.field(k, name)
            if fty is not None:
                return fty, fk
            k = k.outer
        if name in self.jf.static_imports:
            k = self.w.klass(self.jf.static_imports[name])
            if k:
                return self.w.field(k, name)
        return None, None

    def longest_type(self, i, end):
        """Longest prefix of the dotted identifier chain at i that names a class in the sources: (Klass, index after) or (None, i)."""
        t = self.t
        names, j = [t[i][1]], i + 1
        while j + 1 < end and t[j][1] == "." and t[j + 1][0] == "id":
            names.append(t[j + 1][1])
            j += 2
        best, used = None, i
        for n in range(1, len(names) + 1):
            k = self.w.resolve(".".join(names[:n]), self.k)
            if k is None and n > 1:
                k = self.w.klass(".".join(names[:n]))
            if k is not None:
                best, used = k, i + 2 * n - 1
        return best, used

    EXTERNAL_RETURNS = {
        ("String", "length"): "int", ("String", "getBytes"): "byte[]", ("String", "equals"): "boolean", ("String", "trim"): "String",
        ("String", "split"): "String[]", ("List", "size"): "int", ("List", "isEmpty"): "boolean", ("ByteBuffer", "position"): "int",
        ("ByteBuffer", "capacity"): "int", ("ByteBuffer", "putInt"): "ByteBuffer", ("ByteBuffer", "put"): "ByteBuffer",
        ("ByteBuffer", "order"): "ByteBuffer", ("ByteBuffer", "allocateDirect"): "ByteBuffer", ("Math", "max"): None,
        ("Integer", "parseInt"): "int", ("Boolean", "parseBoolean"): "boolean", ("Object", "toString"): "String",
    }

    def external_call(self, cur, name, args):
        base = cur.split("<")[0].split(".")[-1]
        return self.EXTERNAL_RETURNS.get((base, name))

    def call_on(self, k, name, args, implicit=False, static_only=False):
        """Method `name` on class k (a sources class) with argument types; returns the return type name or None."""
        self.last_decl = None
        chain = [k]
        if implicit:
            o = k.outer
            while o is not None:
                chain.append(o)
                o = o.outer
        cands = []
        for c in chain:
            for cc in [c] + self.w.supertypes(c):
                cands += [(m, cc) for m in cc.methods.get(name, [])]
        if not cands:
            if k.kind == "enum" and name in ("values", "valueOf", "name", "ordinal", "equals", "toString", "hashCode", "compareTo"):
                self.c.ok("enum builtin %s" % name)
                return {"values": k.name + "[]", "valueOf": k.name, "name": "String", "ordinal": "int", "equals": "boolean",
                        "toString": "String", "hashCode": "int", "compareTo": "int"}[name]
            if name in ("toString", "equals", "hashCode", "getClass") :
                self.c.ok("Object." + name)
                return {"toString": "String", "equals": "boolean", "hashCode": "int", "getClass": "Class"}[name]
            if any(self.w.external_super(c) for c in chain):
                self.c.external += 1
                return None
            self.c.err(self.jf, "%s has no method `%s` (searched its supertypes in the sources too)" % (k.fq(), name))
            return None
        m, mk, why = self.c.pick(cands, args, self.k)
        if m is None:
            self.c.err(self.jf, "%s.%s(%s): no overload matches (%s); declared: %s" % (
                k.fq(), name, ", ".join(str(a) for a in args), why,
                "; ".join("%s(%s)" % (c.name, ", ".join(p[0] for p in c.params)) for c, _ in cands)))
            return None
        if static_only and "static" not in m.mods and not implicit:
            self.c.err(self.jf, "%s.%s is an instance method, called on the type" % (k.fq(), name))
        self.c.ok("%s.%s/%d" % (mk.fq(), name, len(args)))
        self.last_decl = mk
        ret = m.ret
        if ret in ("?", "void") or ret in getattr(m, "tparams", []) or ret.replace("[]", "") in mk.type_params:
            return None if ret != "void" else "void"
        # the return type is written in mk's context: resolve there and hand back a name usable from here (fq when possible)
        rk = self.w.resolve(ret.replace("[]", ""), mk)
        if rk is not None:
            return rk.fq() + "[]" * ret.count("[]")
        return ret


def check_class(chk, jf, klass, outer_scope=None):
    w = chk.w
    # supertypes exist
    for sup in klass.extends + klass.implements:
        sk = w.resolve(sup, klass if klass.name != "$anon" else klass.outer)
        if sk is None:
            fq = jf.imports.get(sup.split(".")[0], "")
            if fq.startswith("org.apache.pinot.") or sup.startswith("org.apache.pinot."):
                chk.err(jf, "supertype %s of %s not found in the reference" % (sup, klass.name))
            else:
                chk.external += 1
        else:
            chk.ok("supertype %s" % sk.fq())
    # an anonymous class's supertype may be an interface
    if klass.name == "$anon" and klass.extends:
        sk = w.resolve(klass.extends[0], klass.outer)
        if sk is not None and sk.kind == "interface":
            klass.implements, klass.extends = klass.extends, []
    supers = w.supertypes(klass)
    # @Override targets
    for name, ms in klass.methods.items():
        for m in ms:
            if "@Override" in m.mods:
                found = any(len(x.params) == len(m.params) for s in supers for x in s.methods.get(name, []))
                if found:
                    chk.ok("@Override %s.%s/%d" % (klass.name, name, len(m.params)))
                elif w.external_super(klass):
                    chk.external += 1
                else:
                    chk.err(jf, "@Override %s.%s(%s): no supertype in the sources declares it with %d parameters" % (
                        klass.name, name, ", ".join(p[0] for p in m.params), len(m.params)))
    # abstract methods a concrete class must implement
    if klass.kind == "class" and "abstract" not in getattr(klass, "mods", set()):
        concrete = set()
        for c in [klass] + [s for s in supers]:
            for name, ms in c.methods.items():
                for m in ms:
                    if "abstract" not in m.mods:
                        concrete.add((name, len(m.params)))
        for s in supers:
            for name, ms in s.methods.items():
                for m in ms:
                    if "abstract" in m.mods and (name, len(m.params)) not in concrete:
                        chk.err(jf, "%s does not implement %s.%s/%d" % (klass.name if klass.name != "$anon" else "anonymous class", s.fq(), name, len(m.params)))
                    elif "abstract" in m.mods:
                        chk.ok("implements %s.%s/%d" % (s.name, name, len(m.params)))
    # field types, field initialisers are skipped (simple constants here); method bodies
    class_scope = Scope(outer_scope)
    for ms in list(klass.methods.values()) + [klass.ctors]:
        for m in ms:
            bw = BodyWalker(chk, jf, klass, jf.toks)
            sc = Scope(class_scope)
            for ty, pn in m.params:
                sc.vars[pn] = ty
                bw.note_type(ty, 0)
            if m.ret and m.ret not in ("?", "void") and m.name != klass.name:
                bw.note_type(m.ret, 0)
            if m.body:
                bw.walk_block(m.body[0] + 1, m.body[1] - 1, sc)
    for fname, fty in klass.fields.items():
        if klass.kind == "enum" and fname in klass.enum_constants:
            continue
        BodyWalker(chk, jf, klass, jf.toks).note_type(fty, 0)
    for nk in klass.nested.values():
        check_class(chk, jf, nk, None)


def check_file(chk, jf):
    for simple, fq in jf.imports.items():
        if fq.startswith("org.apache.pinot.") and not fq.startswith("org.apache.pinot.gpu."):
            if chk.w.klass(fq) is None:
                chk.err(jf, "import %s: not found in the reference" % fq)
            else:
                chk.ok("import %s" % fq)
        elif not fq.startswith("org.apache.pinot.gpu."):
            chk.external += 1
    for k in jf.classes.values():
        check_class(chk, jf, k)


def run(reference, sources, verbose=False):
    world = World(reference, sources)
    chk = Checker(world, verbose)
    files = []
    for dp, dn, fn in os.walk(sources):
        for f in sorted(fn):
            if f.endswith(".java"):
                files.append(os.path.join(dp, f))
    for p in sorted(files):
        jf = world.files.get(p) or parse_java(p)
        world.files[p] = jf
        check_file(chk, jf)
    return chk, files


def main():
    ap = argparse.ArgumentParser()
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--sources", default=os.path.join(here, "integration", "java"))
    ap.add_argument("-v", action="store_true")
    a = ap.parse_args()
    chk, files = run(a.reference, a.sources, a.v)
    if a.v:
        for d in chk.details:
            print("  ok  " + d)
    print("%d files: %d uses resolved against the reference, %d on external types (java.*, fastutil, clearspring, roaringbitmap), "
          "%d on receivers whose type could not be derived, %d errors" % (len(files), chk.checked, chk.external, chk.untyped, len(chk.errors)))
    for e in chk.errors:
        print("ERROR " + e)
    return 1 if chk.errors else 0


if __name__ == "__main__":
    sys.exit(main())
