"""Heatmaps (reference heatmap.py:40-94).  plotly/kaleido are optional here: when they are not
installed the call logs once and returns None instead of failing the pipeline (visualisation is
outside the accelerated path)."""

import logging

CSS_COLORS = {
    "aliceblue", "antiquewhite", "aqua", "aquamarine", "azure", "beige", "bisque", "black", "blanchedalmond", "blue",
    "blueviolet", "brown", "burlywood", "cadetblue", "chartreuse", "chocolate", "coral", "cornflowerblue", "cornsilk",
    "crimson", "cyan", "darkblue", "darkcyan", "darkgoldenrod", "darkgray", "darkgrey", "darkgreen", "darkkhaki",
    "darkmagenta", "darkolivegreen", "darkorange", "darkorchid", "darkred", "darksalmon", "darkseagreen", "darkslateblue",
    "darkslategray", "darkslategrey", "darkturquoise", "darkviolet", "deeppink", "deepskyblue", "dimgray", "dimgrey",
    "dodgerblue", "firebrick", "floralwhite", "forestgreen", "fuchsia", "gainsboro", "ghostwhite", "gold", "goldenrod",
    "gray", "grey", "green", "greenyellow", "honeydew", "hotpink", "indianred", "indigo", "ivory", "khaki", "lavender",
    "lavenderblush", "lawngreen", "lemonchiffon", "lightblue", "lightcoral", "lightcyan", "lightgoldenrodyellow",
    "lightgray", "lightgrey", "lightgreen", "lightpink", "lightsalmon", "lightseagreen", "lightskyblue", "lightslategray",
    "lightslategrey", "lightsteelblue", "lightyellow", "lime", "limegreen", "linen", "magenta", "maroon",
    "mediumaquamarine", "mediumblue", "mediumorchid", "mediumpurple", "mediumseagreen", "mediumslateblue",
    "mediumspringgreen", "mediumturquoise", "mediumvioletred", "midnightblue", "mintcream", "mistyrose", "moccasin",
    "navajowhite", "navy", "oldlace", "olive", "olivedrab", "orange", "orangered", "orchid", "palegoldenrod", "palegreen",
    "paleturquoise", "palevioletred", "papayawhip", "peachpuff", "peru", "pink", "plum", "powderblue", "purple", "red",
    "rosybrown", "royalblue", "saddlebrown", "salmon", "sandybrown", "seagreen", "seashell", "sienna", "silver", "skyblue",
    "slateblue", "slategray", "slategrey", "snow", "springgreen", "steelblue", "tan", "teal", "thistle", "tomato",
    "turquoise", "violet", "wheat", "white", "whitesmoke", "yellow", "yellowgreen"}

_warned = False


def draw_heatmap(matrix, colors=None, midpoint=0.5, filename=None):
    """Render ``matrix`` as a heatmap to ``filename`` (.html or an image type kaleido supports)."""
    global _warned
    import os
    if os.environ.get("PHAMCLUST_NO_HEATMAPS") == "1":          # visualisation is outside the accelerated path: tests of N = 10,000
        if not _warned:                                          # pipelines do not render 500 cluster heatmaps through kaleido
            logging.info("PHAMCLUST_NO_HEATMAPS=1 - heatmaps are skipped")
            _warned = True
        return None
    try:
        import plotly.express as px
    except ImportError:
        if not _warned:
            logging.warning("plotly is not installed - heatmaps are skipped")
            _warned = True
        return None
    labels = matrix.nodes
    colors = colors or ["red", "yellow", "green"]
    if len(colors) == 2:
        scale = [(0, colors[0]), (1, colors[1])]
    elif len(colors) == 3:
        scale = [(0, colors[0]), (midpoint, colors[1]), (1, colors[2])]
    else:
        raise ValueError(f"expected 2 or 3 colors, got {len(colors)}")
    fig = px.imshow(matrix.to_ndarray(), x=labels, y=labels, color_continuous_scale=scale, zmin=0.0, zmax=1.0)
    if filename is None:
        return fig
    if str(filename).endswith(".html"):
        fig.write_html(filename)
        return filename
    try:
        fig.write_image(filename)
    except Exception as exc:                      # kaleido missing / no headless renderer
        if not _warned:
            logging.warning(f"static heatmap export unavailable ({type(exc).__name__}) - image heatmaps are skipped")
            _warned = True
        return None
    return filename
