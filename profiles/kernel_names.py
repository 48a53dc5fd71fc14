"""How rocprofv3's kernel names are shortened in the profile summaries AND in profiles/pmc_traffic.json, so that the row a number
was read from can be named: every template argument that tells two instantiations apart is kept (NS, dx1, zero_in, cache
policies, no_slip / fuse_grad / reach), never cut off by a length limit (a truncated name had merged the zero_in = true
and zero_in = false kernels into one row)."""


def short(name):
    import re
    # sor_fused_kernel<Lane2<NS, VEC, ZERO_IN[, NT]> | Lane4<NS, ZERO_IN>, NS, DX1, ZERO_IN>
    # (the store policy ST is an int since round 4 -- 0 plain, 2 nt, 16 sc1 -- and was a bool NT before)
    # (rounds 4 - 5 also carried a load policy LD for the chained launch, Lane2<NS, VEC, ZERO_IN, ST, LD>; round 6 retired it and added
    # the arithmetic: Lane2<NS, VEC, ZERO_IN, ST, FOLD>.  Every template argument that tells two instantiations apart is kept: NS,
    # dx1, zero_in, the store policy and the arithmetic -- VERDICT r04: a truncated name had merged the zero_in = true and
    # zero_in = false kernels into one row)
    m = re.search(r"(sor_fused_kernel|sor_chain_kernel)<.*?Lane(\d)<(\d+)((?:, (?:true|false))+)((?:, \d+)*)((?:, (?:true|false))*)>, \d+, (true|false)(?:, (true|false))?>", name)
    if m:
        flags = [f == "true" for f in re.findall(r"true|false", m.group(4))]
        pol = re.findall(r"\d+", m.group(5))
        st = {"2": "nt", "16": "sc1"}.get(pol[0] if pol else "", "nt" if (m.group(2) == "2" and len(flags) >= 3 and flags[2]) else "")
        ld = {"16": "+ldsc1"}.get(pol[1] if len(pol) > 1 else "", "")
        fold = ", fold" if "true" in m.group(6) else ""      # (SFL_OPT_SOR_FOLD = 1; nothing: the reference's two products)
        zero = f", zero_in={m.group(8)}" if m.group(8) else ""
        return f"{m.group(1)}<Lane{m.group(2)}{st}{ld}, NS={m.group(3)}, dx1={m.group(7)}{zero}{fold}>"
    m = re.search(r"(advect_divergence_tiled_kernel|advect_vec2f_tiled_kernel|advect_vec3uq32_tiled_kernel)<([^>]*)>", name)
    if m:
        flags = re.findall(r"true|false", m.group(2))
        label = {"advect_divergence_tiled_kernel": ["no_slip"], "advect_vec2f_tiled_kernel": ["no_slip", "self"],
                 "advect_vec3uq32_tiled_kernel": ["no_slip", "fuse_grad", "reach"]}[m.group(1)]
        return m.group(1) + "<" + ", ".join(f"{a}={b}" for a, b in zip(label, flags)) + ">"
    for key in ("seam_tiled_kernel", "advect_channels_kernel", "copy_bands_kernel", "signal_arrival_kernel"):
        if key in name:
            return key
    for key in ("divergence_tiled_kernel", "gradient_tiled_kernel"):
        if key in name:
            return key
    for key in ("divergence_stream_kernel", "gradient_stream_kernel", "sor_half_sweep_kernel", "advect_vec2f_kernel", "advect_vec3uq32_kernel",
                "divergence_kernel", "subtract_gradient_kernel", "zero_rows_kernel", "apply_forces_kernel"):
        if key in name:
            return key
    # anything else: the demangled name without its argument list, template arguments kept (abbreviated, never cut off in
    # the middle of what distinguishes two kernels)
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"\b(void|sfl::|\(anonymous namespace\)::)", "", name).strip()
    return name if len(name) <= 110 else name[:107] + "..."
