"""Reaction dict (reaction_parser) -> espressopp-shaped objects: one ChemicalReaction, one
FixedPairList + bond interaction per group, one Reaction per equation, type change post-process.
Behaviour of /root/reference/src/chemlab/reaction_setup.py:71-165,408-541 (normal reactions)."""


class SetupReactions(object):
    def __init__(self, espressopp, system, vl, topol, topol_manager, config, args=None):
        self.espp, self.system, self.vl, self.topol, self.tm, self.cfg, self.args = espressopp, system, vl, topol, topol_manager, config, args
        self.name2type = topol.used_atomsym_atomtype
        self.dynamic_types = set()
        self.fpls = []
        self.extensions_to_integrator = []

    def _setup_reaction_normal(self, cr, fpl):
        e, rl = self.espp, cr["reactant_list"]
        n2t = self.name2type
        r_class = e.integrator.RestrictReaction if cr.get("connectivity_map") else e.integrator.Reaction      # reaction_setup.py:75-78
        r = r_class(
            type_1=n2t[rl["type_1"]["name"]], type_2=n2t[rl["type_2"]["name"]],
            delta_1=int(rl["type_1"]["delta"]), delta_2=int(rl["type_2"]["delta"]),
            min_state_1=int(rl["type_1"]["min"]), max_state_1=int(rl["type_1"]["max"]),
            min_state_2=int(rl["type_2"]["min"]), max_state_2=int(rl["type_2"]["max"]),
            rate=float(cr["rate"]), fpl=fpl, cutoff=float(cr["cutoff"]))
        self.dynamic_types.update((r.type_1, r.type_2))
        r.intramolecular = bool(cr["intramolecular"])
        r.intraresidual = bool(cr["intraresidual"])
        r.is_virtual = bool(cr["virtual"])
        if "min_cutoff" in cr:
            r.get_reaction_cutoff().min_cutoff = float(cr["min_cutoff"])
        r.active = cr.get("active", True)
        if cr.get("connectivity_map"):                       # id pairs `b1 b2`, one per line (reaction_setup.py:115-128)
            conn = set()
            with open(cr["connectivity_map"]) as f:
                for line in f:
                    if line.strip():
                        b1, b2 = map(int, line.split())
                        conn.add(tuple(sorted((b1, b2))))
            for b1, b2 in sorted(conn):
                r.define_connection(b1, b2)
        for which in ("type_1", "type_2"):
            old, new = rl[which]["name"], rl[which]["new_type"]
            if old != new:   # type change with the new type's mass and charge from [ atomtypes ]
                pp = e.integrator.PostProcessChangeProperty()
                prop = self.topol.gt.atomtypes[new]
                pp.add_change_property(n2t[old], e.integrator.TopologyParticleProperties(type=n2t[new], mass=prop["mass"], q=prop["charge"]))
                r.add_postprocess(pp, which)
                self.dynamic_types.update((n2t[old], n2t[new]))
        return r

    def _setup_reaction_exchange(self, cr, fpl):
        """`A:B + C -> A:C + B` (reaction_setup.py:167-251): a VIRTUAL A + C reaction -- no bond is made or removed, the event
        only changes properties -- that requires A to carry a bonded B in B's state window; A changes type like a normal
        reactant, the B neighbours of A in the window take B's new type and have their state incremented."""
        e, rl, n2t = self.espp, cr["reactant_list"], self.name2type
        if cr.get("connectivity_map"):
            raise RuntimeError("connectivity_map not supported by exchange reaction")
        rt1, rt2, rt3 = rl["type_1"], rl["type_2"], rl["type_3"]
        r = e.integrator.Reaction(
            type_1=n2t[rt1["name"]], type_2=n2t[rt3["name"]], delta_1=int(rt1["delta"]), delta_2=int(rt3["delta"]),
            min_state_1=int(rt1["min"]), max_state_1=int(rt1["max"]), min_state_2=int(rt3["min"]), max_state_2=int(rt3["max"]),
            rate=float(cr["rate"]), fpl=fpl, cutoff=float(cr["cutoff"]))
        r.is_virtual = True
        r.add_constraint(e.integrator.ReactionConstraintNeighbourState(n2t[rt2["name"]], int(rt2["min"]), int(rt2["max"])), "type_1")
        r.intraresidual = bool(cr["intraresidual"])
        r.intramolecular = bool(cr["intramolecular"])
        if "min_cutoff" in cr:
            r.get_reaction_cutoff().min_cutoff = float(cr["min_cutoff"])
        r.active = cr.get("active", True)
        t1_old, t1_new = n2t[rt1["name"]], n2t[rt1["new_type"]]
        t2_old, t2_new = n2t[rt2["name"]], n2t[rt2["new_type"]]
        self.dynamic_types.update((t1_old, t1_new, t2_old, t2_new, n2t[rt3["name"]]))
        if t1_old != t1_new:
            pp = e.integrator.PostProcessChangePropertyByTopologyManager(self.tm)
            prop = self.topol.gt.atomtypes[rt1["new_type"]]
            pp.add_change_property(t1_old, e.integrator.TopologyParticleProperties(type=t1_new, mass=prop["mass"], q=prop["charge"]))
            r.add_postprocess(pp, "type_1")
        prop = self.topol.gt.atomtypes[rt2["new_type"]]
        tpp = e.integrator.TopologyParticleProperties(type=t2_new, mass=prop["mass"], q=prop["charge"], incr_state=int(rt2["delta"]))
        tpp.set_min_max_state(int(rt2["min"]), int(rt2["max"]))
        nb = e.integrator.PostProcessChangeNeighboursProperty(self.tm)
        nb.add_change_property(t2_old, tpp, 1)
        r.add_postprocess(nb, "type_1")
        return r

    def _setup_extension(self, name):
        """[ext_<name>] -> (post-process object, invoke_on).  In scope: ChangeNeighboursProperty
        (reaction_post_process.py:76-115: `type_transfers=OLD:level->NEW[(state=1,...)],...`)."""
        import re
        cfg = self.cfg["extensions"].get(name)
        if cfg is None:
            raise RuntimeError("extension %s is not defined ([ext_%s] missing)" % (name, name))
        if cfg.get("ext_type") == "ATRPActivator":
            return self._setup_atrp_activator(cfg), None
        if cfg.get("ext_type") != "ChangeNeighboursProperty":
            raise NotImplementedError("reaction extension %s (%s) is outside the hot-path scope (SURVEY f-4)" % (name, cfg.get("ext_type")))
        e, n2t = self.espp, self.name2type
        pp = e.integrator.PostProcessChangeNeighboursProperty(self.tm)
        re_opt = re.compile(r"(?P<type_name>\w+)\(?(?P<options>[a-zA-Z0-9_=,]*)\)?")
        for tr in cfg["type_transfers"].split(","):
            old, new = tr.split("->")
            old_type, nb_level = old.split(":")
            new_type, options = re_opt.match(new).groups()
            prop = self.topol.gt.atomtypes[new_type]
            if "state" not in prop:
                raise RuntimeError("Please define initial atom state in [ atomstate ] section of your topology for atom type %s" % new_type)
            kw = dict(type=n2t[new_type], mass=prop["mass"], q=prop["charge"], state=prop["state"])
            for kv in filter(None, options.split(",")):
                k, v = kv.split("=")
                kw[k] = int(v) if k in ("state", "type") else float(v)
            pp.add_change_property(n2t[old_type], e.integrator.TopologyParticleProperties(**kw), int(nb_level))
            self.dynamic_types.update((n2t[old_type], n2t[new_type]))
        return pp, cfg.get("invoke_on", "both")

    def _setup_atrp_activator(self, cfg):
        """[ext_*] ext_type=ATRPActivator -> integrator extension (reaction_post_process.py:380-426):
        options=`TYPE(state,A|DA)->NEWTYPE(delta);...`, flag DA marks the centres that react with the deactivator."""
        import re
        e, n2t = self.espp, self.name2type
        out_prefix = getattr(self.args, "output_prefix", "sim") if self.args is not None else "sim"
        seed = getattr(self.args, "rng_seed", 0) if self.args is not None else 0
        act = e.integrator.ATRPActivator(self.system, int(cfg["interval"]), int(cfg["num_particles"]), float(cfg["ratio_activator"]),
                                         float(cfg["ratio_deactivator"]), float(cfg["delta_catalyst"]), float(cfg["k_activate"]),
                                         float(cfg["k_deactivate"]))
        act.stats_filename = cfg.get("stats_file", "%s_%s_atrp_stats.dat" % (out_prefix, seed))
        act.select_from_all = int(cfg.get("select_from_all", 1))
        re_reactant = re.compile(r"(?P<name>\w+)\((?P<state>\d+),\s*(?P<flag>[AD]{1,2})\)")
        re_product = re.compile(r"(?P<new_type>\w+)\((?P<delta>[0-9-]+)\)")
        for opt in cfg["options"].split(";"):
            to_process, after_process = opt.split("->")
            reactant = re_reactant.match(to_process.strip()).groupdict()
            product = re_product.match(after_process.strip()).groupdict()
            if reactant["flag"] not in ("A", "DA"):
                raise RuntimeError('Flag %s not "A" or "DA"' % reactant["flag"])
            prop = self.topol.gt.atomtypes[product["new_type"]]
            act.add_reactive_center(type_id=n2t[reactant["name"]], state=int(reactant["state"]), is_activator=reactant["flag"] == "DA",
                                    new_property=e.integrator.TopologyParticleProperties(type=n2t[product["new_type"]], mass=prop["mass"], q=prop["charge"]),
                                    delta_state=int(product["delta"]))
            self.dynamic_types.update((n2t[reactant["name"]], n2t[product["new_type"]]))
        return act

    def setup_reactions(self):
        e, g = self.espp, self.cfg["general"]
        ar = e.integrator.ChemicalReaction(self.system, self.vl, self.system.storage, self.tm, g["interval"])
        ar.nearest_mode = g["nearest"]
        if g["max_per_interval"] > 0:
            ar.max_per_interval = g["max_per_interval"]
        for gname, group in self.cfg["reactions"].items():
            group_ext = [(name, self._setup_extension(name)) for name in group["extensions"]]
            # integrator extensions (ATRPActivator) go to the driver, post-processes to the group's reactions (reaction_setup.py:470-483)
            self.extensions_to_integrator.extend(x for _, (x, inv) in group_ext if inv is None)
            group_pp = [(name, v) for name, v in group_ext if v[1] is not None]
            fpl = e.FixedPairList(self.system.storage)
            pot_class = getattr(e.interaction, group["potential"])
            pot = pot_class(**group["potential_options"])
            inter_class = getattr(e.interaction, "FixedPairList%s" % group["potential"])
            inter = inter_class(self.system, fpl, pot)
            self.system.addInteraction(inter, "fpl_%s" % gname)
            self.fpls.append((gname, fpl, inter))
            fpl.type_list = set()        # (type pairs, before and after, of the group's reactions: FPLDef.type_list, reaction_setup.py:436,532)
            for cr in group["reaction_list"]:
                cr["connectivity_map"] = group.get("connectivity_map")      # group level -> reaction level (reaction_setup.py:488)
                r = self._setup_reaction_exchange(cr, fpl) if cr.get("reaction_type") == "exchange" else self._setup_reaction_normal(cr, fpl)
                for name, (pp, invoke_on) in group_pp:        # reaction_setup.py:495-505
                    if name not in cr.get("exclude_extensions", []):
                        r.add_postprocess(pp, invoke_on)
                ar.add_reaction(r)
                n2t, rl = self.name2type, cr["reactant_list"]
                second = "type_3" if cr.get("reaction_type") == "exchange" else "type_2"
                fpl.type_list.add((n2t[rl["type_1"]["name"]], n2t[rl[second]["name"]]))
                fpl.type_list.add((n2t[rl["type_1"]["new_type"]], n2t[rl[second]["new_type"]]))
        return ar, self.fpls
