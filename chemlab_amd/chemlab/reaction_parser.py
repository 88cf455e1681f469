"""Reaction .cfg (INI) -> dict, normal (bond-forming / virtual) reactions only.
Grammar and defaults of /root/reference/src/chemlab/reaction_parser.py:36-66,130-232,235-266
(SURVEY.md Appendix A).  Dissociation / exchange equations and `ext_*` extensions are out of scope
and raise NotImplementedError when a reaction actually needs them."""
import ast
import collections
import configparser
import re

REACTION_NORMAL = "normal"

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
    try:
        data["reactant_list"], data["reaction_type"] = parse_equation(sec["reaction"])
    except ValueError as e:
        raise NotImplementedError("only normal reactions A(a,b) + B(c,d) -> A'(x):B'(y) are in scope: %s" % e)
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
