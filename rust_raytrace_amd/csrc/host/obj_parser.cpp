// obj_parser.cpp — restatement of raytrace_lib/src/obj_parser.rs:20-73.
// "v x y z" and "f a[/..] b c ..." lines only, 1-based indices, first three
// corners of a face; everything else is ignored.  Where the reference panics
// (unwrap / assert / index out of bounds) this throws std::runtime_error.
//
// ObjMode::Robust is an opt-in EXTENSION the reference does not have (its
// parser would panic or silently drop geometry on such files): faces with more
// than three corners are fan-triangulated (c0, c_i, c_i+1), negative indices
// count back from the vertices read so far (OBJ convention), comment and blank
// lines inside face data are tolerated, degenerate triangles are skipped
// instead of aborting the load.  The default stays the reference's behaviour.
#include <cerrno>
#include <cstdlib>
#include <fstream>
#include <sstream>

#include "raytrace.hpp"

namespace raytrace {
namespace obj_parser {

namespace {
// str::split_whitespace
std::vector<std::string> fields(const std::string& s) {
    std::vector<std::string> out;
    size_t i = 0;
    while (i < s.size()) {
        while (i < s.size() && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < s.size() && !isspace((unsigned char)s[j])) j++;
        if (j > i) out.push_back(s.substr(i, j - i));
        i = j;
    }
    return out;
}
float parse_f32(const std::string& tok, const std::string& line) {
    char* end = nullptr;
    errno = 0;
    const float x = strtof(tok.c_str(), &end);  // correctly rounded decimal -> f32, like Rust's parse::<f32>
    if (end == tok.c_str() || *end != '\0') throw std::runtime_error("parse_obj: bad number in vertex line: " + line);
    return x;
}
size_t parse_index(const std::string& tok, const std::string& line) {
    const std::string head = tok.substr(0, tok.find('/'));  // x.split('/').nth(0)
    if (head.empty() || head.find_first_not_of("0123456789") != std::string::npos)
        throw std::runtime_error("parse_obj: bad index in face line: " + line);
    return (size_t)strtoull(head.c_str(), nullptr, 10);
}
// Robust mode: "-k" is the k-th last vertex read so far; returns a 1-based absolute index (0 = invalid)
size_t parse_index_relative(const std::string& tok, const std::string& line, size_t nverts) {
    const std::string head = tok.substr(0, tok.find('/'));
    if (head.size() > 1 && head[0] == '-' && head.find_first_not_of("0123456789", 1) == std::string::npos) {
        const size_t back = (size_t)strtoull(head.c_str() + 1, nullptr, 10);
        return back >= 1 && back <= nverts ? nverts - back + 1 : 0;
    }
    return parse_index(tok, line);
}
}  // namespace

std::vector<Triangle> parse_obj(const std::string& path, const Vec3& offset, float scale,
                                const std::tuple<Vec3, Vec3, Vec3>& transform, const SurfaceKind& surface,
                                float edge_thickness, ObjMode mode) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("parse_obj: cannot read " + path);
    std::vector<Vec3> vertices;
    std::vector<std::vector<size_t>> faces;
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.compare(0, 2, "v ") == 0) {
            const auto parts = fields(line.substr(2));
            if (parts.size() != 3) throw std::runtime_error("parse_obj: vertex line needs exactly 3 numbers: " + line);
            vertices.push_back(make_vec(parse_f32(parts[0], line), parse_f32(parts[1], line), parse_f32(parts[2], line)));
        } else if (line.compare(0, 2, "f ") == 0) {
            std::vector<size_t> corners;
            for (const auto& tok : fields(line.substr(2)))
                corners.push_back(mode == ObjMode::Robust ? parse_index_relative(tok, line, vertices.size()) : parse_index(tok, line));
            faces.push_back(std::move(corners));
        }
    }
    std::vector<Triangle> objs;
    objs.reserve(faces.size());
    auto place = [&](size_t idx) {  // obj_parser.rs:61-70
        if (idx < 1 || idx > vertices.size()) throw std::runtime_error("parse_obj: face index out of range");
        return vertices[idx - 1].mult(scale).change_basis(transform).add(offset);
    };
    for (const auto& face : faces) {
        if (face.size() < 3) throw std::runtime_error("parse_obj: face with fewer than 3 corners");
        if (mode == ObjMode::Reference) {  // first three corners, whatever follows (obj_parser.rs:63-65)
            Vec3 pts[3] = {place(face[0]), place(face[1]), place(face[2])};
            objs.push_back(make_triangle(pts, surface, edge_thickness));
            continue;
        }
        for (size_t k = 1; k + 1 < face.size(); k++) {  // fan: (c0, ck, ck+1)
            Vec3 pts[3] = {place(face[0]), place(face[k]), place(face[k + 1])};
            try { objs.push_back(make_triangle(pts, surface, edge_thickness)); }
            catch (const std::runtime_error&) {}  // degenerate sliver of a polygon: skipped (the reference would panic, raytrace.rs:357)
        }
    }
    return objs;
}

}  // namespace obj_parser
}  // namespace raytrace
