//! `plonk::ConstraintSystem<Fr>` -> `dehalo_constraint_system` (include/dehalo.h): the walk a patched `keygen_pk` / `ProvingKey::read` performs once per circuit.
//! Executable versions of the same walk: `native.ConstraintSystemDescriptor` (Python, delay-encryption-in-halo2_amd/native.py) and `halo2_amd::ConstraintSystem`
//! (C++, host/halo2_backend.hpp); the library's tests pin both against the CPU restatement's statement of the MainGate + RangeChip shape (oracle/shapes.py).
//!
//! `Expression::{Constant, Fixed, Advice, Instance, Negated, Sum, Product, Scaled}` map 1:1 onto `DEHALO_EXPR_*` nodes, children before parents.  `Selector` does not
//! occur after `compress_selectors`; `Challenge` does not occur in the reference's one-phase circuits (src/lib.rs:164-318).
use dehalo_sys as sys;
use halo2_proofs::plonk::{Any, Column, ConstraintSystem, Expression};
use halo2_proofs::poly::Rotation;
use halo2curves::bn256::Fr;

/// Owns the arrays a `dehalo_constraint_system` points into.
pub struct Descriptor {
    nodes: Vec<sys::dehalo_expr_node>,
    constants: Vec<Fr>,
    gates: Vec<u32>,
    lookup_lens: Vec<u32>,
    lookup_inputs: Vec<u32>,
    lookup_tables: Vec<u32>,
    permutation_columns: Vec<sys::dehalo_column_query>,
    advice_queries: Vec<sys::dehalo_column_query>,
    fixed_queries: Vec<sys::dehalo_column_query>,
    instance_queries: Vec<sys::dehalo_column_query>,
    num_advice: u32,
    num_fixed: u32,
    num_instance: u32,
    minimum_degree: u32,
}

fn column_kind(any: &Any) -> u32 {
    match any {
        Any::Advice(_) => sys::DEHALO_COLUMN_ADVICE as u32,
        Any::Fixed => sys::DEHALO_COLUMN_FIXED as u32,
        Any::Instance => sys::DEHALO_COLUMN_INSTANCE as u32,
    }
}

fn query(col: &Column<Any>, rot: Rotation) -> sys::dehalo_column_query {
    sys::dehalo_column_query { kind: column_kind(col.column_type()), index: col.index() as u32, rotation: rot.0 }
}

impl Descriptor {
    pub fn from_constraint_system(cs: &ConstraintSystem<Fr>) -> Self {
        let mut d = Descriptor {
            nodes: vec![], constants: vec![], gates: vec![], lookup_lens: vec![], lookup_inputs: vec![], lookup_tables: vec![],
            permutation_columns: cs.permutation().get_columns().iter().map(|c| query(c, Rotation::cur())).collect(),
            advice_queries: cs.advice_queries().iter().map(|(c, r)| sys::dehalo_column_query { kind: sys::DEHALO_COLUMN_ADVICE as u32, index: c.index() as u32, rotation: r.0 }).collect(),
            fixed_queries: cs.fixed_queries().iter().map(|(c, r)| sys::dehalo_column_query { kind: sys::DEHALO_COLUMN_FIXED as u32, index: c.index() as u32, rotation: r.0 }).collect(),
            instance_queries: cs.instance_queries().iter().map(|(c, r)| sys::dehalo_column_query { kind: sys::DEHALO_COLUMN_INSTANCE as u32, index: c.index() as u32, rotation: r.0 }).collect(),
            num_advice: cs.num_advice_columns() as u32,
            num_fixed: cs.num_fixed_columns() as u32,
            num_instance: cs.num_instance_columns() as u32,
            minimum_degree: cs.minimum_degree().unwrap_or(0) as u32,
        };
        for gate in cs.gates() {
            for poly in gate.polynomials() {
                let root = d.lower(poly);
                d.gates.push(root);
            }
        }
        for lookup in cs.lookups() {
            d.lookup_lens.push(lookup.input_expressions().len() as u32);
            for e in lookup.input_expressions() {
                let r = d.lower(e);
                d.lookup_inputs.push(r);
            }
            for e in lookup.table_expressions() {
                let r = d.lower(e);
                d.lookup_tables.push(r);
            }
        }
        d
    }

    fn push(&mut self, kind: u32, a: u32, b: u32, rotation: i32) -> u32 {
        self.nodes.push(sys::dehalo_expr_node { kind, a, b, rotation });
        (self.nodes.len() - 1) as u32
    }

    fn constant(&mut self, c: Fr) -> u32 {
        if let Some(i) = self.constants.iter().position(|x| *x == c) {
            return i as u32;
        }
        self.constants.push(c);
        (self.constants.len() - 1) as u32
    }

    /// post-order: children are pushed (and numbered) before their parent
    fn lower(&mut self, e: &Expression<Fr>) -> u32 {
        match e {
            Expression::Constant(c) => {
                let i = self.constant(*c);
                self.push(sys::DEHALO_EXPR_CONSTANT as u32, i, 0, 0)
            }
            Expression::Fixed(q) => self.push(sys::DEHALO_EXPR_FIXED as u32, q.column_index() as u32, 0, q.rotation().0),
            Expression::Advice(q) => self.push(sys::DEHALO_EXPR_ADVICE as u32, q.column_index() as u32, 0, q.rotation().0),
            Expression::Instance(q) => self.push(sys::DEHALO_EXPR_INSTANCE as u32, q.column_index() as u32, 0, q.rotation().0),
            Expression::Negated(a) => {
                let a = self.lower(a);
                self.push(sys::DEHALO_EXPR_NEGATED as u32, a, 0, 0)
            }
            Expression::Sum(a, b) => {
                let (a, b) = (self.lower(a), self.lower(b));
                self.push(sys::DEHALO_EXPR_SUM as u32, a, b, 0)
            }
            Expression::Product(a, b) => {
                let (a, b) = (self.lower(a), self.lower(b));
                self.push(sys::DEHALO_EXPR_PRODUCT as u32, a, b, 0)
            }
            Expression::Scaled(a, c) => {
                let a = self.lower(a);
                let i = self.constant(*c);
                self.push(sys::DEHALO_EXPR_SCALED as u32, a, i, 0)
            }
            Expression::Selector(_) => panic!("dehalo: selectors must be compressed into fixed columns first (keygen does)"),
            Expression::Challenge(_) => panic!("dehalo: multi-phase circuits are outside the reference's shapes"),
        }
    }

    /// The C view.  Valid while `self` is alive and unmodified.
    pub fn as_c(&self) -> sys::dehalo_constraint_system {
        sys::dehalo_constraint_system {
            num_advice: self.num_advice, num_fixed: self.num_fixed, num_instance: self.num_instance, minimum_degree: self.minimum_degree,
            nodes: self.nodes.as_ptr(), num_nodes: self.nodes.len() as u32,
            constants: self.constants.as_ptr() as *const u64, num_constants: self.constants.len() as u32,
            gates: self.gates.as_ptr(), num_gates: self.gates.len() as u32,
            lookup_lens: self.lookup_lens.as_ptr(), num_lookups: self.lookup_lens.len() as u32,
            lookup_inputs: self.lookup_inputs.as_ptr(), lookup_tables: self.lookup_tables.as_ptr(),
            permutation_columns: self.permutation_columns.as_ptr(), num_permutation_columns: self.permutation_columns.len() as u32,
            advice_queries: self.advice_queries.as_ptr(), num_advice_queries: self.advice_queries.len() as u32,
            fixed_queries: self.fixed_queries.as_ptr(), num_fixed_queries: self.fixed_queries.len() as u32,
            instance_queries: self.instance_queries.as_ptr(), num_instance_queries: self.instance_queries.len() as u32,
        }
    }
}
