"""Reaction dict (reaction_parser) -> espressopp-shaped objects: one ChemicalReaction, one
FixedPairList + bond interaction per group, one Reaction per equation, type change post-process.
Behaviour of /root/reference/src/chemlab/reaction_setup.py:71-165,408-541 (normal reactions)."""


class SetupReactions(object):
    def __init__(self, espressopp, system, vl, topol, topol_manager, config, args=None):
        self.espp, self.system, self.vl, self.topol, self.tm, self.cfg, self.args = espressopp, system, vl, topol, topol_manager, config, args
        self.name2type = topol.used_atomsym_atomtype
        self.dynamic_types = set()
        self.fpls = []

    def _setup_reaction_normal(self, cr, fpl):
        e, rl = self.espp, cr["reactant_list"]
        n2t = self.name2type
        r = e.integrator.Reaction(
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
        for which in ("type_1", "type_2"):
            old, new = rl[which]["name"], rl[which]["new_type"]
            if old != new:   # type change with the new type's mass and charge from [ atomtypes ]
                pp = e.integrator.PostProcessChangeProperty()
                prop = self.topol.gt.atomtypes[new]
                pp.add_change_property(n2t[old], e.integrator.TopologyParticleProperties(type=n2t[new], mass=prop["mass"], q=prop["charge"]))
                r.add_postprocess(pp, which)
                self.dynamic_types.update((n2t[old], n2t[new]))
        return r

    def setup_reactions(self):
        e, g = self.espp, self.cfg["general"]
        ar = e.integrator.ChemicalReaction(self.system, self.vl, self.system.storage, self.tm, g["interval"])
        ar.nearest_mode = g["nearest"]
        if g["max_per_interval"] > 0:
            ar.max_per_interval = g["max_per_interval"]
        for gname, group in self.cfg["reactions"].items():
            if group["extensions"]:
                raise NotImplementedError("reaction extensions %s (ATRPActivator, ChangeNeighboursProperty, ...) are outside the hot-path scope (SURVEY f-4)" % group["extensions"])
            fpl = e.FixedPairList(self.system.storage)
            pot_class = getattr(e.interaction, group["potential"])
            pot = pot_class(**group["potential_options"])
            inter_class = getattr(e.interaction, "FixedPairList%s" % group["potential"])
            inter = inter_class(self.system, fpl, pot)
            self.system.addInteraction(inter, "fpl_%s" % gname)
            self.fpls.append((gname, fpl, inter))
            for cr in group["reaction_list"]:
                ar.add_reaction(self._setup_reaction_normal(cr, fpl))
        return ar, self.fpls
