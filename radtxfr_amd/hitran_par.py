"""HITRAN `.par` / `.data` line files -> the column dict the line-sum consumes.

The reference reads `<table>.data|.par` + `<table>.header` pairs into LOCAL_TABLE_CACHE through hapi's storage layer
(misc/hapi.py:1535-1672): storage2cache() walks every line with getRowObjectFromString(), which cuts the fixed-width
fields named by header['order'] with the printf widths of header['format'] (regex FORMAT_PYTHON_REGEX, :1453),
converts 'd' fields with int(), 'e'/'f' fields with float(), keeps 's' fields verbatim, then reads the optional
comma-separated header['extra'] parameters; A ROW THAT FAILS ANY CONVERSION IS SILENTLY SKIPPED (the bare
`except: continue`, :1650-1654). A `.par` file without a header gets HITRAN_DEFAULT_HEADER (:492-559; createHeader
:1710). This module reproduces exactly that behaviour (pinned by tests/golden/g12_*.npz, columns captured from the
reference's own db_begin() on the committed tests/golden/g12_*.par files):

  * the isotopologue field is '%1d': '0' parses to local_iso_id 0 -- the reference's ISO / TIPS tables are keyed
    (M, 0) for the tenth isotopologue (misc/hapi.py:3388) -- and the letter codes 'A', 'B' of newer HITRAN editions
    fail int() and drop the row, as in the reference;
  * a record whose g' / g'' fields (columns 147-160, '%7.1f') are blank, or that is shorter than 160 characters, is
    dropped for the same reason; blank lines too.

Full-width rows are parsed with NumPy slicing (no per-line Python loop); anything irregular falls back to a per-row
path that follows getRowObjectFromString statement by statement.
"""
import json
import os
import re

import numpy as np

# HITRAN_DEFAULT_HEADER (misc/hapi.py:492-559): field order and printf formats of the 160-character record
DEFAULT_ORDER = ("molec_id", "local_iso_id", "nu", "sw", "a", "gamma_air", "gamma_self", "elower", "n_air", "delta_air",
                 "global_upper_quanta", "global_lower_quanta", "local_upper_quanta", "local_lower_quanta", "ierr", "iref",
                 "line_mixing_flag", "gp", "gpp")
DEFAULT_FORMAT = {
    "molec_id": "%2d", "local_iso_id": "%1d", "nu": "%12.6f", "sw": "%10.3E", "a": "%10.3E", "gamma_air": "%5.4f",
    "gamma_self": "%5.3f", "elower": "%10.4f", "n_air": "%4.2f", "delta_air": "%8.6f", "global_upper_quanta": "%15s",
    "global_lower_quanta": "%15s", "local_upper_quanta": "%15s", "local_lower_quanta": "%15s", "ierr": "%6s",
    "iref": "%12s", "line_mixing_flag": "%1s", "gp": "%7.1f", "gpp": "%7.1f"}
DEFAULT_DEFAULT = {
    "a": 0.0, "gamma_air": 0.0, "gp": "FFF", "local_iso_id": 0, "molec_id": 0, "sw": 0.0, "local_lower_quanta": "000",
    "local_upper_quanta": "000", "gpp": "FFF", "elower": 0.0, "n_air": 0.0, "delta_air": 0.0, "global_upper_quanta": "000",
    "iref": "EEE", "line_mixing_flag": "EEE", "ierr": "EEE", "nu": 0.0, "gamma_self": 0.0, "global_lower_quanta": "000"}
_FORMAT_RE = re.compile(r"^%(\d*)(\.(\d*))?([edfsEDFS])$")  # FORMAT_PYTHON_REGEX, misc/hapi.py:1453


def default_header(table_name="###"):
    return {"table_type": "column-fixed", "size_in_bytes": -1, "table_name": table_name, "number_of_rows": -1,
            "order": list(DEFAULT_ORDER), "format": dict(DEFAULT_FORMAT), "default": dict(DEFAULT_DEFAULT)}


def _fields(order, fmt):
    """[(name, width, kind)] with kind 'd' (int), 'f' (float), 's' (string) or '?' (a format the reference's
    getRowObjectFromString raises on -- every row is then dropped)."""
    out = []
    for name in order:
        m = _FORMAT_RE.search(fmt[name])
        if m is None:
            raise Exception('Format "%s" is unknown' % fmt[name])  # re.search(...).groups() on None in the reference
        lng, _, _, ty = m.groups()
        kind = "d" if ty == "d" else "f" if ty.lower() in ("e", "f") else "s" if ty == "s" else "?"
        out.append((name, int(lng), kind))
    return out


def _parse_row(line, fixed, extra, sep, has_order):
    """getRowObjectFromString (misc/hapi.py:1535-1589) for one line (trailing newline included, as the reference's
    `for line in InfileData` hands it over). Raises like the reference; the caller drops the row."""
    vals = []
    pos = 0
    for _, lng, kind in fixed:
        s = line[pos:pos + lng]
        if kind == "d":
            vals.append(int(s))
        elif kind == "f":
            vals.append(float(s))
        elif kind == "s":
            vals.append(s)
        else:
            raise Exception("unknown format")
        pos += lng
    if extra:
        chunks = line.split(sep)
        pos = 1 if has_order else 0
        for _, _, kind in extra:
            s = chunks[pos]  # IndexError drops the row, as in the reference
            if kind == "d":
                try:
                    v = int(s)
                except Exception:
                    v = 0
            elif kind == "f":
                try:
                    v = float(s)
                except Exception:
                    v = 0.0
            elif kind == "s":
                v = s
            else:
                raise Exception("unknown format")
            vals.append(v)
            pos += 1
    return vals


def read_header(path):
    """The JSON header next to a data file (`<base>.header`), or None."""
    hp = os.path.splitext(path)[0] + ".header"
    if not os.path.isfile(hp):
        return None
    with open(hp, "r") as f:
        text = f.read()
    try:
        return json.loads(text)
    except Exception:
        raise Exception("Invalid header")


def read_table(path, header=None):
    """Parse `path` the way the reference's storage2cache does. Returns (header, columns): numeric columns are NumPy
    int64 / float64 arrays, string columns lists of str; header['number_of_rows'] is the count of rows that parsed.
    `header` defaults to the sibling `.header` file, else HITRAN_DEFAULT_HEADER."""
    if header is None:
        header = read_header(path) or default_header(os.path.splitext(os.path.basename(path))[0])
    header = json.loads(json.dumps(header))  # private deep copy
    order = list(header.get("order", []))
    extra_names = list(header.get("extra", []))
    both = set(order) & set(extra_names)
    if both:
        raise Exception("Parameters with the same names: {}".format(both))
    fixed = _fields(order, header.get("format", {})) if order else []
    extra = _fields(extra_names, header.get("extra_format", {})) if extra_names else []
    sep = header.get("extra_separator", ",")
    width = sum(w for _, w, _ in fixed)
    with open(path, "r") as f:  # text mode: universal newlines, like the reference
        lines = f.readlines()
    names = order + extra_names
    kinds = [k for _, _, k in fixed + extra]
    cols = None
    regular = (not extra) and bool(fixed) and all(k != "?" for k in kinds)
    if regular and lines:
        # vectorised path: every line at least `width` characters before its newline
        body = [ln[:-1] if ln.endswith("\n") else ln for ln in lines]
        if all(len(b) >= width for b in body):
            try:
                block = np.array([b[:width] for b in body], dtype="U%d" % width)
                chars = block.view("U1").reshape(len(body), width)
                cols = {}
                pos = 0
                for name, lng, kind in fixed:
                    text = np.ascontiguousarray(chars[:, pos:pos + lng]).view("U%d" % lng)[:, 0]
                    if kind == "d":
                        cols[name] = text.astype(np.int64)  # element-wise int(): raises on 'A', '  '
                    elif kind == "f":
                        cols[name] = text.astype(np.float64)
                    else:
                        cols[name] = text.tolist()
                    pos += lng
            except ValueError:
                cols = None  # some row does not parse: take the exact per-row path and drop it there
    if cols is None:
        rows = []
        for ln in lines:
            try:
                rows.append(_parse_row(ln, fixed, extra, sep, bool(order)))
            except Exception:
                continue
        cols = {}
        for c, (name, kind) in enumerate(zip(names, kinds)):
            v = [r[c] for r in rows]
            cols[name] = np.asarray(v, dtype=np.int64) if kind == "d" else np.asarray(v, dtype=np.float64) if kind == "f" else v
    n = len(cols[names[0]]) if names else 0
    # storage2cache folds the comma-separated parameters into the fixed ones (:1656-1667)
    fmt = dict(header.get("format", {}))
    fmt.update(header.get("extra_format", {}))
    for k in ("extra", "extra_format", "extra_separator"):
        header.pop(k, None)
    header["order"] = names
    header["format"] = fmt
    header["number_of_rows"] = n
    return header, cols


def read_par(path, header=None):
    """Columns of a `.par` / `.data` file as the reference's storage2cache leaves them in
    LOCAL_TABLE_CACHE[name]['data'] (see read_table)."""
    return read_table(path, header)[1]


def _fit(text, width):
    """HITRAN drops the zero before the decimal point when a value does not fit its field ('-.005000', '.0712')."""
    if len(text) > width:
        text = text.replace("0.", ".", 1)
    if len(text) != width:
        raise ValueError("value %r does not fit a %d-character .par field" % (text, width))
    return text


def write_par(path, columns):
    """Write a column dict as full 160-character `.par` records that the reference's parser (and read_par) accept:
    the ten numeric line parameters from `columns`, quantum-number / reference fields from `columns` when present
    (else blank), g' / g'' as '%7.1f' (0.0 when absent -- left blank the reference drops the row)."""
    n = len(columns["nu"])

    def col(name, default):
        return columns[name] if name in columns else [default] * n

    a, gp, gpp = col("a", 0.0), col("gp", 0.0), col("gpp", 0.0)
    strs = {k: col(k, "") for k in DEFAULT_ORDER[10:17]}
    with open(path, "w") as f:
        for r in range(n):
            iso = int(columns["local_iso_id"][r])
            if not 0 <= iso <= 9:
                raise ValueError("local_iso_id %d does not fit the '%%1d' field (the tenth isotopologue is 0, "
                                 "misc/hapi.py:3388)" % iso)
            rec = ("%2d" % int(columns["molec_id"][r]) + "%1d" % iso
                   + _fit("%12.6f" % columns["nu"][r], 12) + _fit("%10.3E" % columns["sw"][r], 10) + _fit("%10.3E" % a[r], 10)
                   + _fit("%5.4f" % columns["gamma_air"][r], 5) + _fit("%5.3f" % columns["gamma_self"][r], 5)
                   + _fit("%10.4f" % columns["elower"][r], 10) + _fit("%4.2f" % columns["n_air"][r], 4)
                   + _fit("%8.6f" % columns["delta_air"][r], 8))
            for k in DEFAULT_ORDER[10:17]:
                w = int(_FORMAT_RE.search(DEFAULT_FORMAT[k]).group(1))
                rec += (DEFAULT_FORMAT[k] % str(strs[k][r]))[:w]
            rec += _fit("%7.1f" % float(gp[r]), 7) + _fit("%7.1f" % float(gpp[r]), 7)
            assert len(rec) == 160, (len(rec), rec)
            f.write(rec + "\n")


def storage2cache(TableName, path):
    """Load `path` into radtxfr_amd.hapi.LOCAL_TABLE_CACHE[TableName] (misc/hapi.py:1615-1672): the header (order,
    format, number_of_rows) + the data columns. Returns the columns."""
    from . import hapi
    header, cols = read_table(path)
    header["table_name"] = header.get("table_name", TableName)
    hapi.LOCAL_TABLE_CACHE[TableName] = {"header": header, "data": cols}
    print("                     Lines parsed: %d" % header["number_of_rows"])
    return cols
