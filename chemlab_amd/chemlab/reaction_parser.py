"""Reaction .cfg (INI) -> dict, normal (bond-forming / virtual) reactions only.
Grammar and defaults of /root/reference/src/chemlab/reaction_parser.py:36-66,130-232,235-266
(SURVEY.md Appendix A).  Dissociation / exchange equations and `ext_*` extensions are out of scope
and raise NotImplementedError when a reaction actually needs them."""
import ast
import collections
import configparser
import re

REACTION_NORMAL = "normal"
REACTION_EXCHANGE = "exchange"

_RE_REACTANT = re.compile(r"(?P<name>\w+)\((?P<min>\d+),\s*(?P<max>\d+)\)")
_RE_PRODUCT = re.compile(r"(?P<name>\w+)\((?P<delta>[0-9-]+)\)")


def parse_equation(text):
    """`T1(min,max) + T2(min,max) -> N1(d1):N2(d2)`; the state window is half-open [min,max),
    d is an increment; a type change happens iff product name != reactant name."""
    reactants, products = text.split("->")
    parts = [p.strip() for p in reactants.split("+")]
    if len(parts) != 2:
        raise ValueError("normal reaction needs two reactants: %r" % text)
    a, b = (_RE_REACTANT.match(p) for p in parts)
    if a is None or b is None or ":" in parts[0] or ":" in parts[1]:
        raise ValueError("cannot parse reactants of %r" % text)
    prods = [_RE_PRODUCT.match(p.strip()) for p in products.split(":")]
    if len(prods) != 2 or None in prods or "+" in products:
        raise ValueError("cannot parse products of %r" % text)
    rl = {"type_1": a.groupdict(), "type_2": b.groupdict()}
    for k, p in zip(("type_1", "type_2"), prods):
        rl[k]["delta"] = p.group("delta")
        rl[k]["new_type"] = p.group("name")
    return rl, REACTION_NORMAL


def parse_exchange_equation(text):
    """`A(a,b):B(c,d) + C(e,f) -> A'(dA):C'(dC) + B'(dB)` (reaction_parser.py:98-127): type_1 = A, type_2 = B (the leaving
    group, takes the product behind the `+`), type_3 = C (takes the second product of the `:` pair)."""
    re_prod = re.compile(r"(?P<new_type>\w+)\((?P<delta>[0-9-]+)\)")
    reactants, products = (t.strip() for t in text.split("->"))
    part_a, part_b = [[p.strip() for p in x.split(":")] for x in reactants.split("+")]
    mol_a, mol_b = (_RE_REACTANT.match(p).groupdict() for p in part_a)
    mol_c = _RE_REACTANT.match(part_b[0]).groupdict()
    prod_a, prod_b = [[p.strip() for p in x.split(":")] for x in products.split("+")]
    pa, pb = (re_prod.match(p).groupdict() for p in prod_a)
    pc = re_prod.match(prod_b[0]).groupdict()
    mol_a.update(pa); mol_b.update(pc); mol_c.update(pb)
    return {"type_1": mol_a, "type_2": mol_b, "type_3": mol_c}, REACTION_EXCHANGE


def _literal(v, default=False):
    if v is None:
        return default
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def process_reaction(sec):
    sec = dict(sec)
    data = {"rate": float(sec["rate"]), "intramolecular": _literal(sec.get("intramolecular")),
            "intraresidual": _literal(sec.get("intraresidual")), "virtual": _literal(sec.get("virtual")),
            "exclude_extensions": set(), "equation": sec["reaction"], "active": _literal(sec.get("active"), True)}
    if "exclude_extensions" in sec:
        data["exclude_extensions"] = {s.strip() for s in sec["exclude_extensions"].split(",")}
    eq = sec["reaction"]
    left = eq.split("->")[0]
    if ":" in left and "+" in left:             # A:B + C -> A:C + B  (reaction_parser.py:98-127)
        data["reactant_list"], data["reaction_type"] = parse_exchange_equation(eq)
    elif ":" in left:                           # A:B -> A + B: dissociation needs BasicDynamicResolution (AdResS), out of scope
        raise NotImplementedError("dissociation reactions A(a,b):B(c,d) -> A'(x) + B'(y) need the dynamic-resolution machinery (reaction_setup.py:253-356), out of scope")
    else:
        try:
            data["reactant_list"], data["reaction_type"] = parse_equation(eq)
        except ValueError as e:
            raise NotImplementedError("cannot parse reaction %r: %s" % (eq, e))
    if "min_cutoff" in sec:
        data["min_cutoff"] = float(sec["min_cutoff"])
    if "sigma" in sec and "eq_distance" in sec:
        raise NotImplementedError("smooth reaction cutoff (sigma/eq_distance) is outside the hot-path scope")
    if "cutoff" not in sec:
        raise RuntimeError("Please define cutoff of the reaction: %s" % sec["reaction"])
    data["cutoff"] = float(sec["cutoff"])
    return sec["group"], data


def process_general(sec):
    sec = dict(sec)
    return {"interval": int(sec["interval"]),
            # quirk Q1: bool('0') is True -- `nearest=0` ENABLES nearest-partner mode
            "nearest": bool(sec.get("nearest", False)),
            "pair_distances_filename": sec.get("pair_distances_filename"),
            "max_per_interval": int(sec.get("max_per_interval", -1))}


def process_group(sec):
    sec = dict(sec)
    opts = {}
    for kv in sec.get("potential_options", "").split(","):
        if kv.strip():
            k, v = kv.split("=")
            opts[k.strip()] = _literal(v.strip())
    ext = [e.strip() for e in sec.get("extensions", "").split(",") if e.strip()]
    return {"potential": sec["potential"], "potential_options": opts, "extensions": ext,
            "connectivity_map": sec.get("connectivity_map"), "reaction_list": []}


def parse_config(file_name):
    """-> {'general': {...}, 'reactions': {group: {...}}, 'extensions': {...}} in file order."""
    cp = configparser.RawConfigParser(delimiters=(":", "="), comment_prefixes=("#", ";"), inline_comment_prefixes=None)
    cp.optionxform = str
    cp.read(file_name)
    cfg = {"general": process_general(cp.items("general")), "reactions": collections.OrderedDict(), "extensions": {}}
    for s in cp.sections():
        if s.startswith("group_"):
            cfg["reactions"][s[len("group_"):]] = process_group(cp.items(s))
        elif s.startswith("ext_"):
            cfg["extensions"][s[len("ext_"):]] = dict(cp.items(s))
        elif s.startswith("reaction_"):
            group, data = process_reaction(cp.items(s))
            data["name"] = s[len("reaction_"):]
            cfg["reactions"][group]["reaction_list"].append(data)
    return cfg
