"""A small C# declaration scanner (test infrastructure): the public surface of the classes in a .cs file.

Not a C# parser: it strips comments / strings, finds `class Name` blocks and lists, at class-body depth, the members a
caller (or Unity's serializer / message dispatch) can see:
  * public constructors, methods, properties, indexers, fields          -> "ctor(...)", "Type Name(...)", "Type Name", ...
  * [SerializeField] fields (Unity serialises them by name and type)     -> "serialized Type name"
  * Unity message methods by name (Awake, Update, ... are found by reflection whatever their access level)
Parameter lists are reduced to their types.  Used by tests/test_csharp_surface.py and tests/golden/make_csharp_surface.py."""
import re

UNITY_MESSAGES = {"Awake", "Start", "Update", "OnRenderImage", "OnDestroy", "OnDrawGizmos", "OnEnable", "OnDisable"}


MODIFIERS = {"public", "private", "protected", "internal", "static", "override", "virtual", "readonly", "sealed", "unsafe", "extern"}


def _strip(src):
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r'\$?"(?:\\.|[^"\\])*"', '""', src)
    return src


def _param_types(params):
    params = params.strip()
    if not params:
        return ""
    out, depth, cur = [], 0, ""
    for ch in params:
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    types = []
    for p in out:
        p = re.sub(r"^(?:\[\w[^\]]*\]\s*)+", "", p.strip())    # leading attributes such as [Out]
        p = p.split("=")[0].strip()                          # default values
        toks = p.split()
        types.append(" ".join(toks[:-1]) if len(toks) > 1 else toks[0])
    return ", ".join(types)


def _class_bodies(src):
    for m in re.finditer(r"\b(?:public\s+)?(?:static\s+|sealed\s+|abstract\s+)*class\s+(\w+)\s*(<[^>{]*>)?([^{]*)\{", src):
        start, depth, i = m.end(), 1, m.end()
        while depth and i < len(src):
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        yield m.group(1), (m.group(2) or "").replace(" ", ""), src[start:i - 1]


def _depth0(body):
    """the class body with nested { } blocks blanked out (member bodies), attributes kept"""
    out, depth = [], 0
    for ch in body:
        if ch == "{":
            depth += 1
            out.append("{" if depth == 1 else " ")
        elif ch == "}":
            out.append("}" if depth == 1 else " ")
            depth -= 1
        else:
            out.append(ch if depth == 0 else " ")
    return "".join(out)


def surface(path_or_text, is_text=False):
    src = _strip(path_or_text if is_text else open(path_or_text, encoding="utf-8-sig").read())
    result = {}
    for name, generics, body in _class_bodies(src):
        flat = _depth0(body)
        members = set()
        # constructors, methods, indexers (declarations followed by a body, `=>` or `;`)
        head = (r"((?:\[[^\]]*\]\s*)*)((?:public|private|protected|internal|static|override|virtual|readonly|sealed|unsafe|extern)\s+)*"
                r"([\w<>\[\],\.\s]*?)\b(\w+)\s*")
        tail = r"\s*(?::\s*this\s*\([^)]*\)\s*)?(?=\{|=>|;)"
        decls = [(m, "(") for m in re.finditer(head + r"\(([^()]*(?:\([^()]*\)[^()]*)*)\)" + tail, flat)]
        decls += [(m, "[") for m in re.finditer(head + r"\[([^\[\]]*)\]" + tail, flat)]
        for m, opener in decls:
            mods = flat[m.start():m.start(4)]
            rtype, mname, params = m.group(3).strip(), m.group(4), m.group(5)
            rtype = " ".join(t for t in rtype.split() if t not in MODIFIERS)
            if re.search(r"\b(if|for|while|switch|return|new|using|lock|foreach|catch)\b", mname):
                continue
            is_public = re.search(r"\bpublic\b", mods) is not None
            if opener == "[":
                if mname == "this" and is_public:
                    members.add(f"{rtype} this[{_param_types(params)}]")
                continue
            if mname == name and not rtype:
                if is_public:
                    members.add(f"ctor({_param_types(params)})")
            elif rtype and mname in UNITY_MESSAGES and rtype.split()[-1] == "void":
                members.add(f"message {mname}({_param_types(params)})")
            elif rtype and is_public:
                members.add(f"{rtype} {mname}({_param_types(params)})")
        # properties and fields: `[attrs] mods Type Name` followed by =>, {, = or ;
        for m in re.finditer(r"((?:\[[^\]]*\]\s*)*)((?:(?:public|private|protected|internal|static|readonly|const)\s+)+)([\w<>\[\],\.]+)\s+(\w+)\s*(=>|\{|=|;)", flat):
            attrs, mods, ftype, fname = m.group(1), m.group(2), m.group(3), m.group(4)
            if "SerializeField" in attrs:
                members.add(f"serialized {ftype} {fname}")
            elif re.search(r"\bpublic\b", mods):
                members.add(f"{ftype} {fname}")
        result[name + generics] = sorted(members)
    return result
