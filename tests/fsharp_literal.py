"""A third restatement of the reference's sampling path: a line-by-line Python transliteration of the F# source, kept as close
to the original's SHAPE as Python allows (recursive bestCandidate, mutable Ray/LightRay objects, ValueOption as None, the
while loops with isDone flags).  It exists only to cross-check oracle/oracle.cpp, which was written in a different style;
pure Python, so only tiny images.  Citations: /root/reference/RayTracing/<file>:<line>.

Python floats are IEEE doubles and CPython never fuses a*b+c, so every operation rounds as .NET's does.  `** 5.0` is evaluated
exactly (fractions) and rounded once -- the same definition of Math.Pow(x, 5.0) the oracle and the HIP path use.
"""
import math
from fractions import Fraction

TOL = 0.00000001  # Float.fs:82
NAN, INF = float("nan"), float("inf")


def fsqrt(x):  # F# `sqrt` of a negative (or NaN) is NaN; math.sqrt raises instead
    return math.sqrt(x) if x >= 0.0 else NAN


def f_equal(a, b):  # Float.fs:84
    return abs(a - b) < TOL


def f_positive(a):  # Float.fs:88
    return a > TOL


def f_compare(a, b):  # Float.fs:90-96
    if abs(a - b) < TOL:
        return "Equal"
    elif a < b:
        return "Less"
    else:
        return "Greater"


class FloatProducer:  # Float.fs:14-76
    def __init__(self, x, y, z, w):
        self.x, self.y, self.z, self.w = x, y, z, w

    def _gen(self):  # generateInt32, Float.fs:14-20
        t = (self.x ^ ((self.x << 11) & 0xFFFFFFFF)) & 0xFFFFFFFF
        self.x = self.y
        self.y = self.z
        self.z = self.w
        self.w = (self.w ^ (self.w >> 19) ^ (t ^ (t >> 8))) & 0xFFFFFFFF
        return self.w

    @staticmethod
    def _to_double(w):  # toInt + toDouble, Float.fs:22-29
        i = ((w & 0xFF) << 24) ^ (((w >> 8) & 0xFF) << 16) ^ (((w >> 16) & 0xFF) << 8) ^ ((w >> 24) & 0xFF)
        return float(i) / float(4294967295)

    def Get(self):
        return self._to_double(self._gen())

    def GetTwo(self):
        one, two = self._gen(), self._gen()
        return self._to_double(one), self._to_double(two)

    def GetThree(self):
        one, two, three = self._gen(), self._gen(), self._gen()
        return self._to_double(one), self._to_double(two), self._to_double(three)


# ---- Point.fs ---------------------------------------------------------------------------------------------------------
def v_dot(p, q):  # Point.fs:18
    return p[0] * q[0] + p[1] * q[1] + p[2] * q[2]


def v_scale(s, v):  # Point.fs:22-24
    return (s * v[0], s * v[1], s * v[2])


def v_diff(p, q):  # Point.fs:26 / Point.differenceToThenFrom Point.fs:94
    return (p[0] - q[0], p[1] - q[1], p[2] - q[2])


def v_unitise(vec):  # Point.fs:28-35
    dot = v_dot(vec, vec)
    if f_equal(dot, 0.0):
        return None
    factor = 1.0 / fsqrt(dot)
    return v_scale(factor, vec)


def uv_random(fp):  # Point.fs:49-59 (recursion)
    r1, r2, r3 = fp.GetThree()
    x, y, z = (2.0 * r1) - 1.0, (2.0 * r2) - 1.0, (2.0 * r3) - 1.0
    res = v_unitise((x, y, z))
    return uv_random(fp) if res is None else res


# ---- Ray.fs (a heap class with mutable fields) ----------------------------------------------------------------------------
class Ray:
    def __init__(self, origin, vector):
        self.Origin, self.Vector = origin, vector


def ray_overwrite_with_make(origin, vector, ray):  # Ray.fs:11-24
    dot = v_dot(vector, vector)
    if f_equal(dot, 0.0):
        return False
    ray.Origin = origin
    factor = 1.0 / fsqrt(dot)
    ray.Vector = v_scale(factor, vector)
    return True


def ray_make_prime(origin, vector):  # Ray.fs:26-34
    v = v_unitise(vector)
    return None if v is None else Ray(origin, v)


def walk_along_ray(o, v, magnitude):  # Ray.fs:42-43
    return (o[0] + (v[0] * magnitude), o[1] + (v[1] * magnitude), o[2] + (v[2] * magnitude))


def walk_along(ray, magnitude):  # Ray.fs:45-46
    return walk_along_ray(ray.Origin, ray.Vector, magnitude)


def flip_in_place(r):  # Ray.fs:79-81
    r.Vector = v_scale(-1.0, r.Vector)


# ---- Pixel.fs ---------------------------------------------------------------------------------------------------------
def round_half_even_byte(x):  # Math.Round |> byte
    return int(round(x)) & 0xFF  # Python's round() is banker's rounding, like Math.Round


def pixel_combine(p1, p2):  # Pixel.fs:136-141
    return ((p1[0] * p2[0]) // 255, (p1[1] * p2[1]) // 255, (p1[2] * p2[2]) // 255)


def pixel_darken(albedo, p):  # Pixel.fs:144-151
    return (round_half_even_byte(float(p[0]) * albedo), round_half_even_byte(float(p[1]) * albedo), round_half_even_byte(float(p[2]) * albedo))


# ---- Plane.fs:64-79 -----------------------------------------------------------------------------------------------------
def make_orthonormal_spanned_by(r1, r2):
    coefficient = v_dot(r1.Vector, r2.Vector)
    vec2 = v_unitise(v_diff(r2.Vector, v_scale(coefficient, r1.Vector)))
    if vec2 is None:
        return None
    return (r1.Vector, vec2, r1.Origin)  # V1, V2, Point


def v_sum(a, b):  # Point.fs:20
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def v_cross(p, q):  # Point.fs:44-45
    (x, y, z), (a, b, c) = p, q
    return (y * c - z * b, z * a - x * c, x * b - a * y)


def make_normal_to(point, v):  # Plane.fs:22-36
    x, y, z = v
    v1 = (0.0, 0.0, 1.0) if f_equal(z, 0.0) else (1.0, 1.0, (-x - y) / z)
    v2 = v_unitise(v_cross(v, v1))
    v1 = v_unitise(v1)
    return (v1, v2, point)


def plane_basis(viewUp, plane):  # Plane.fs:82-98
    V1, V2, P = plane
    viewUp = v_unitise(viewUp)
    v1Component = v_dot(V1, viewUp)
    v2Component = v_dot(V2, viewUp)
    v2 = v_unitise(v_sum(v_scale(v1Component, V1), v_scale(v2Component, V2)))
    v1 = v_unitise(v_sum(v_scale(v2Component, V1), v_scale(-v1Component, V2)))
    return Ray(P, v1), Ray(P, v2)


def camera_make_basic(samplesPerPixel, focalLength, aspectRatio, origin, viewDirection, viewUp):  # Camera.fs:34-59
    height = 2.0
    view = Ray(origin, viewDirection)
    corner = walk_along(view, focalLength)
    viewPlane = make_normal_to(corner, viewDirection)
    xAxis, yAxis = plane_basis(viewUp, viewPlane)
    return {"eye": view.Origin, "view": view.Vector, "xo": xAxis.Origin, "xd": xAxis.Vector, "yo": yAxis.Origin, "yd": yAxis.Vector,
            "vw": aspectRatio * height, "vh": height, "focal": focalLength, "spp": samplesPerPixel, "depth": 150}


# ---- BoundingBox.fs:25-94 ---------------------------------------------------------------------------------------------
def inverse_directions(ray):
    def inv(c):
        try:
            return 1.0 / c
        except ZeroDivisionError:
            return math.copysign(INF, c)
    return (inv(ray.Vector[0]), inv(ray.Vector[1]), inv(ray.Vector[2]))


def bbox_hits(inv, ray, box):
    (invX, invY, invZ), (x, y, z), (mn, mx) = inv, ray.Origin, box
    tMin, tMax = -INF, INF

    def mul(a, b):  # IEEE: 0 * inf = NaN (Python gives nan as well)
        return a * b

    t0, t1 = mul(mn[0] - x, invX), mul(mx[0] - x, invX)
    if invX < 0.0:
        t0, t1 = t1, t0
    tMin = t0 if t0 > tMin else tMin
    tMax = t1 if t1 < tMax else tMax
    if tMax < tMin or 0.0 >= tMax:
        return False
    t0, t1 = mul(mn[1] - y, invY), mul(mx[1] - y, invY)
    if invY < 0.0:
        t0, t1 = t1, t0
    tMin = t0 if t0 > tMin else tMin
    tMax = t1 if t1 < tMax else tMax
    if tMax < tMin or 0.0 >= tMax:
        return False
    t0, t1 = mul(mn[2] - z, invZ), mul(mx[2] - z, invZ)
    if invZ < 0.0:
        t0, t1 = t1, t0
    tMin = t0 if t0 > tMin else tMin
    tMax = t1 if t1 < tMax else tMax
    return tMax >= tMin and tMax >= 0.0


# ---- Texture.fs -------------------------------------------------------------------------------------------------------
# Texture = ("Colour", pixel) | ("Arbitrary", point -> pixel); ParameterisedTexture = ("Colour", pixel) |
# ("Checkered", even, odd, gridSize) | ("Image", rows) | ("Arbitrary", x -> y -> Texture), closures as in F#.
# Math.Acos / Math.Atan2 / Math.Sin are the C runtime's in .NET; the path defines them as the correctly rounded values
# (csrc/rt_trig.h).  Third route, after the device's double-double and the oracle's binary128: mpmath at 250 bits, rounded once.
# Zeros, infinities and NaNs (exact in every libm) go to math.*.
import mpmath as _mp

_mp.mp.prec = 250


PLATFORM_LIBM = False  # True: Math.Acos / Atan2 / Sin are this platform's libm (what a real .NET run calls: within an ulp of the
                       # correctly rounded value, not always equal to it) -- test_platform_libm_textures_informational


def _plain(*xs):
    return (not PLATFORM_LIBM) and all(x == x and abs(x) != math.inf and x != 0.0 for x in xs)


def facos(x):  # F# `acos` outside [-1, 1] is NaN; math.acos raises instead
    if not -1.0 <= x <= 1.0:
        return NAN
    return float(_mp.acos(_mp.mpf(x))) if _plain(x) else math.acos(x)


def fatan2(y, x):
    return float(_mp.atan2(_mp.mpf(y), _mp.mpf(x))) if _plain(y, x) else math.atan2(y, x)


def fsin(x):
    return float(_mp.sin(_mp.mpf(x))) if _plain(x) else math.sin(x)


def plane_map_inverse(radius, centre):  # Sphere.fs:55-61, curried like the reference's use of it
    def interpret(p):
        x, y, z = v_scale(1.0 / radius, v_diff(p, centre))
        theta = facos(-y)
        phi = fatan2(-z, x) + math.pi
        return (phi / (2.0 * math.pi)), theta / math.pi
    return interpret


def texture_colour_at(point, t):  # Texture.fs:12-15
    return t[1] if t[0] == "Colour" else t[1](point)


def param_colour_at(interpret, t, p):  # Texture.fs:50-67
    if t[0] == "Colour":
        return t[1]
    if t[0] == "Arbitrary":
        x, y = interpret(p)
        return texture_colour_at(p, t[1](x, y))
    if t[0] == "Checkered":
        _, even, odd, gridSize = t
        x, y = interpret(p)
        sine = fsin(gridSize * x) * fsin(gridSize * y)
        if f_compare(sine, 0.0) == "Less":
            return param_colour_at(interpret, even, p)
        return param_colour_at(interpret, odd, p)
    if t[0] == "Image":
        img = t[1]
        x, y = interpret(p)
        x = int((1.0 - x) * float(len(img[0]) - 1))
        y = int(y * float(len(img) - 1))
        return img[y][x]
    raise ValueError(t[0])


def to_texture(interpret, texture):  # Texture.fs:69-72
    if texture[0] == "Colour":
        return ("Colour", texture[1])
    return ("Arbitrary", lambda p: param_colour_at(interpret, texture, p))


# ---- Sphere.fs ---------------------------------------------------------------------------------------------------------
def pow5(x):  # Math.Pow (x, 5.0): exact fifth power, rounded once
    if not math.isfinite(x):
        return x * x * x * x * x
    return float(Fraction(x) ** 5)


def reflect_without_fuzz(normal, strike, light):  # Sphere.fs:68-87
    plane = make_orthonormal_spanned_by(normal, light["Ray"])
    if plane is None:
        flip_in_place(light["Ray"])
        light["Ray"].Origin = strike
    else:
        V1, V2, P = plane
        normalComponent = -v_dot(V1, light["Ray"].Vector)
        tangentComponent = v_dot(V2, light["Ray"].Vector)
        dest = walk_along_ray(walk_along_ray(P, V1, normalComponent), V2, tangentComponent)
        ray_overwrite_with_make(strike, v_diff(dest, strike), light["Ray"])


def add_fuzz(fuzz, rand, strike, reflected):  # Sphere.fs:89-104
    isDone = False
    while not isDone:
        offset = uv_random(rand)
        sphereCentre = walk_along(reflected["Ray"], 1.0)
        target = walk_along_ray(sphereCentre, offset, fuzz)
        isDone = ray_overwrite_with_make(strike, v_diff(target, strike), reflected["Ray"])


def refract(inside, normal, strike, incomingCos, index, light):  # Sphere.fs:108-146
    index = 1.0 / index if inside else index / 1.0
    plane = make_orthonormal_spanned_by(normal, light["Ray"])
    if plane is None:
        ray_overwrite_with_make(strike, light["Ray"].Vector, light["Ray"])
        return
    incomingSin = fsqrt(1.0 - incomingCos * incomingCos)
    outgoingSin = incomingSin / index
    if f_compare(outgoingSin, 1.0) == "Greater":
        reflect_without_fuzz(normal, strike, light)
        return
    outgoingCos = fsqrt(1.0 - outgoingSin * outgoingSin)
    outgoingPoint = walk_along(Ray(walk_along(normal, -outgoingCos), plane[1]), outgoingSin)
    ray_overwrite_with_make(strike, v_diff(outgoingPoint, strike), light["Ray"])


def sphere_reflection(s, light, strike, rand):  # Sphere.fs:150-300; returns a Pixel (absorbed) or None
    centre, radius, r2 = s["centre"], s["radius"], s["radius"] * s["radius"]
    flipped = f_compare(radius, 0.0) == "Less"  # Sphere.fs:321
    inside = False
    normal = ray_make_prime(strike, v_diff(strike, centre))  # Sphere.normal, Sphere.fs:65-66
    d = v_diff(centre, light["Ray"].Origin)
    c = f_compare(v_dot(d, d), r2)
    if c in ("Equal", "Less"):
        if not flipped:
            inside = True
            flip_in_place(normal)
    else:
        if flipped:
            inside = True
            flip_in_place(normal)
    style = s["style"]
    tex = texture_colour_at(strike, s["tex"]) if "tex" in s else s["rgb"]  # LightSourceCap carries a Pixel, not a Texture
    if style == "LightSource":
        return pixel_combine(light["Colour"], tex)
    if style == "LightSourceCap":
        lower = centre[0] + (radius - (radius / 4.0))
        return pixel_combine(tex, light["Colour"]) if f_compare(strike[0], lower) == "Greater" else (0, 0, 0)
    newColour = pixel_darken(s["albedo"], pixel_combine(light["Colour"], tex))
    if style == "Lambert":
        light["Colour"] = newColour
        sphereCentre = walk_along(normal, 1.0)
        isDone = False
        while not isDone:
            offset = uv_random(rand)
            target = walk_along_ray(sphereCentre, offset, 1.0)
            isDone = ray_overwrite_with_make(strike, v_diff(target, strike), light["Ray"])
        return None
    if style == "Pure":
        reflect_without_fuzz(normal, strike, light)
        light["Colour"] = newColour
        return None
    if style == "Fuzzed":
        light["Colour"] = newColour
        reflect_without_fuzz(normal, strike, light)
        add_fuzz(s["fuzz"], rand, strike, light)
        return None
    if style == "Dielectric":
        r = rand.Get()
        if r > s["prob"]:
            light["Colour"] = newColour
            reflect_without_fuzz(normal, strike, light)
        else:
            incomingCos = v_dot(light["Ray"].Vector, normal.Vector)
            refract(inside, normal, strike, incomingCos, s["ior"], light)
            light["Colour"] = newColour
        return None
    if style == "Glass":
        incomingCos = v_dot(v_scale(-1.0, light["Ray"].Vector), normal.Vector)
        r = rand.Get()
        sr = 1.0 / s["ior"] if inside else s["ior"]
        param = (1.0 - sr) / (1.0 + sr)
        param = param * param
        reflectionProb = param + (1.0 - param) * pow5(1.0 - incomingCos)
        if r < reflectionProb:
            reflect_without_fuzz(normal, strike, light)
        else:
            refract(inside, normal, strike, incomingCos, s["ior"], light)
        light["Colour"] = newColour
        return None
    raise ValueError(style)


def sphere_first_intersection(s, ray):  # Sphere.fs:349-386
    difference = v_diff(ray.Origin, s["centre"])
    b = v_dot(ray.Vector, difference)
    c = v_dot(difference, difference) - s["radius"] * s["radius"]
    disc = b * b - c
    cmp = f_compare(disc, 0.0)
    if cmp == "Equal":
        ip = -b
    elif cmp == "Less":
        ip = None
    else:
        intermediate = fsqrt(disc)
        i1, i2 = intermediate - b, -(b + intermediate)
        i1Pos, i2Pos = f_positive(i1), f_positive(i2)
        if i1Pos and i2Pos:
            ip = i2 if f_compare(i1, i2) == "Greater" else i1
        elif i1Pos:
            ip = i1
        elif i2Pos:
            ip = i2
        else:
            ip = None
    if ip is None:
        return None
    return ip if f_positive(ip) else None


# ---- InfinitePlane.fs ---------------------------------------------------------------------------------------------------
def plane_pure_outgoing(strike, normal, incoming):  # InfinitePlane.fs:18-38
    plane = make_orthonormal_spanned_by(Ray(strike, normal), incoming)
    if plane is None:
        return Ray(strike, v_scale(-1.0, incoming.Vector))
    V1, V2, P = plane
    normalComponent = -(v_dot(V1, incoming.Vector))
    tangentComponent = v_dot(V2, incoming.Vector)
    s = walk_along(Ray(walk_along(Ray(P, V1), normalComponent), V2), tangentComponent)
    return ray_make_prime(strike, v_diff(s, strike))


def plane_reflection(pl, light, strike, rand):  # InfinitePlane.fs:43-99
    style = pl["style"]
    if style == "LightSource":
        return pixel_combine(light["Colour"], texture_colour_at(strike, pl["tex"]) if "tex" in pl else pl["rgb"])
    newColour = pixel_darken(pl["albedo"], pixel_combine(light["Colour"], pl["rgb"]))
    if style == "Fuzzed":
        pure = plane_pure_outgoing(strike, pl["normal"], light["Ray"])
        outgoing = None
        while outgoing is None:
            offset = uv_random(rand)
            sphereCentre = walk_along(pure, 1.0)
            target = walk_along(Ray(sphereCentre, offset), pl["fuzz"])
            outgoing = ray_make_prime(strike, v_diff(target, strike))
        light["Colour"] = newColour
        light["Ray"] = outgoing
        return None
    if style == "Lambert":
        sphereCentre = walk_along(Ray(strike, pl["normal"]), 1.0)
        offset = uv_random(rand)
        target = walk_along(Ray(sphereCentre, offset), 1.0)
        outgoing = ray_make_prime(strike, v_diff(target, strike))
        if outgoing is None:
            return (0, 0, 0)  # ValueOption.get would throw (DESIGN.md "Undefined cases")
        light["Colour"] = newColour
        light["Ray"] = outgoing
        return None
    if style == "Pure":
        light["Colour"] = newColour
        light["Ray"] = plane_pure_outgoing(strike, pl["normal"], light["Ray"])
        return None
    raise ValueError(style)


def plane_intersection(pl, ray):  # InfinitePlane.fs:125-136
    denominator = v_dot(pl["normal"], ray.Vector)
    if f_equal(denominator, 0.0):
        return None
    t = v_dot(pl["normal"], v_diff(pl["point"], ray.Origin)) / denominator
    return t if f_positive(t) else None


# ---- Hittable.fs / BoundingBoxTree.fs / Scene.fs -------------------------------------------------------------------------------
def hittable_hits(ray, h):  # Hittable.fs:27-31
    return plane_intersection(h, ray) if h["kind"] == "plane" else sphere_first_intersection(h, ray)


def sphere_box(s):  # Sphere.make, Sphere.fs:333-336
    c, r = s["centre"], s["radius"]
    return ((c[0] + (-r), c[1] + (-r), c[2] + (-r)), (c[0] + r, c[1] + r, c[2] + r))


def merge_two(i, j):  # BoundingBox.fs:96-108
    return (tuple(min(a, b) for a, b in zip(i[0], j[0])), tuple(max(a, b) for a, b in zip(i[1], j[1])))


def box_volume(b):  # BoundingBox.fs:13-16
    return (b[1][0] - b[0][0]) * (b[1][1] - b[0][1]) * (b[1][2] - b[0][2])


def tree_make(boxes):  # BoundingBoxTree.fs:9-43; sorted() is stable, as DESIGN.md "Tree shape" defines
    if not boxes:
        return None

    def go(boxes):
        boundAll = boxes[0][1]
        for _, b in boxes[1:]:
            boundAll = merge_two(boundAll, b)
        if len(boxes) == 1:
            return ("Leaf", boxes[0][0], boxes[0][1])
        if len(boxes) == 2:
            return ("Branch", ("Leaf", boxes[0][0], boxes[0][1]), ("Leaf", boxes[1][0], boxes[1][1]), boundAll)
        choices = []
        for axis in range(3):
            srt = sorted(boxes, key=lambda hb: hb[1][0][axis])
            left, right = srt[: len(srt) // 2 + 1], srt[len(srt) // 2 + 1:]
            lb, rb = left[0][1], right[0][1]
            for _, b in left[1:]:
                lb = merge_two(lb, b)
            for _, b in right[1:]:
                rb = merge_two(rb, b)
            choices.append((box_volume(lb) + box_volume(rb), left, right))
        best = choices[0]
        for c in choices[1:]:
            if c[0] < best[0]:
                best = c
        return ("Branch", go(best[1]), go(best[2]), boundAll)

    return go(boxes)


def best_candidate(inv, ray, bestFloat, bestObject, bestLength, box):  # Scene.fs:30-60
    if box[0] == "Leaf":
        _, obj, b = box
        if bbox_hits(inv, ray, b):
            point = hittable_hits(ray, obj)
            if point is None:
                return bestFloat, bestObject, bestLength
            a = point * point
            if a < bestFloat:
                return a, obj, point
            return bestFloat, bestObject, bestLength
        return bestFloat, bestObject, bestLength
    _, left, right, allb = box
    if bbox_hits(inv, ray, allb):
        bestFloat, bestObject, bestLength = best_candidate(inv, ray, bestFloat, bestObject, bestLength, left)
        return best_candidate(inv, ray, bestFloat, bestObject, bestLength, right)
    return bestFloat, bestObject, bestLength


def scene_make(objects):  # Scene.fs:15-28
    bounded = [(h, sphere_box(h)) for h in objects if h["kind"] == "sphere"]  # Hittable.BoundingBox, Hittable.fs:14-18
    unbounded = [h for h in objects if h["kind"] != "sphere"]  # "usphere" and "plane"
    return {"UnboundedObjects": unbounded, "BoundingBoxes": tree_make(bounded)}


def hit_object(s, ray):  # Scene.fs:62-91
    best, bestLength, bestFloat = None, NAN, INF
    if s["BoundingBoxes"] is not None:
        bestFloat, best, bestLength = best_candidate(inverse_directions(ray), ray, bestFloat, best, bestLength, s["BoundingBoxes"])
    for i in s["UnboundedObjects"]:
        point = hittable_hits(ray, i)
        if point is not None:
            a = point * point
            if f_compare(a, bestFloat) == "Less":
                bestFloat, best, bestLength = a, i, point
    if math.isnan(bestLength):
        return None
    return best, walk_along(ray, bestLength)


def trace_ray(maxCount, scene, light, rand):  # Scene.fs:93-114
    bounces, result, isDone = 0, (0, 0, 0), False
    while bounces <= maxCount and not isDone:
        things = hit_object(scene, light["Ray"])
        if things is None:
            isDone = True
        else:
            obj, strike = things
            stop = plane_reflection(obj, light, strike, rand) if obj["kind"] == "plane" else sphere_reflection(obj, light, strike, rand)
            if stop is not None:
                isDone, result = True, stop
            else:
                bounces += 1
    return (205, 105, 180) if not isDone else result


def trace_once(scene, rand, cam, maxW, maxH, row, col, stats):  # Scene.fs:118-155
    r1, r2 = rand.GetTwo()
    landingPoint = ((float(col) + r1) * cam["vw"]) / float(maxW)
    pointOnXAxis = walk_along(Ray(cam["xo"], cam["xd"]), landingPoint)
    walkDistance = ((float(row) + r2) * cam["vh"]) / float(maxH)
    endPoint = walk_along_ray(pointOnXAxis, cam["yd"], walkDistance)
    ray = ray_make_prime(cam["eye"], v_diff(endPoint, cam["eye"]))
    if ray is None:
        result = (0, 0, 0)
    else:
        light = {"Ray": ray, "Colour": (255, 255, 255)}
        result = trace_ray(cam["depth"], scene, light, rand)
    stats[0] += 1
    stats[1] += result[0]
    stats[2] += result[1]
    stats[3] += result[2]


def render_pixel(scene, stream_for, pixel_index, cam, maxW, maxH, row, col):  # Scene.fs:157-194
    stats = [0, 0, 0, 0]
    sample = [0]

    def once():
        trace_once(scene, stream_for(pixel_index, sample[0]), cam, maxW, maxH, row, col, stats)
        sample[0] += 1

    firstTrial = min(5, cam["spp"] // 2)
    for _ in range(0, firstTrial + 1):
        once()
    oldMean = (stats[1] // stats[0], stats[2] // stats[0], stats[3] // stats[0])
    for _ in range(1, firstTrial + 1):
        once()
    newMean = (stats[1] // stats[0], stats[2] // stats[0], stats[3] // stats[0])
    difference = sum(abs(a - b) for a, b in zip(newMean, oldMean))
    if difference != 0:
        for _ in range(1, cam["spp"] - 2 * firstTrial - 1 + 1):
            once()
    return stats


def render(scene, stream_for, cam, maxW, maxH):  # Scene.fs:196-236
    rowsIter, colsIter = 2 * maxH + 1, 2 * maxW + 1
    out = []
    for r in range(rowsIter):
        row = maxH - r - 1
        out.append([render_pixel(scene, stream_for, r * colsIter + c, cam, maxW, maxH, row, c - maxW) for c in range(colsIter)])
    return out


# ---- Seeding: this build's definition, not the reference's (which draws from System.Random) ----------------
M64 = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15


def mix64(z):
    z &= M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def make_stream_for(seed):
    K = mix64(seed + GOLDEN)

    def stream_for(pixel, sample):
        hp = mix64(K ^ ((pixel * 0xD1B54A32D192ED03 + 0x8CB92BA72F3D8DD7) & M64))
        a = mix64(hp + (2 * sample + 1) * GOLDEN)
        b = mix64(hp + (2 * sample + 2) * GOLDEN)
        x, y, z, w = (a & 0xFFFFFFFF) % 2147483647, (a >> 32) % 2147483647, (b & 0xFFFFFFFF) % 2147483647, (b >> 32) % 2147483647
        if x == 0 and y == 0 and z == 0 and w == 0:
            w = 1
        return FloatProducer(x, y, z, w)

    return stream_for
