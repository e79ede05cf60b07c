! racgpu_mod.f90 -- ISO_C_BINDING view of include/racgpu.h plus the host-side mirror of the reference's
! chemistry interface for the per-cell solve.
!
! A Fortran host (rac-2d's disk.f90-style cell sweep) uses this module instead of `module chemistry` for the
! hot path: it keeps the `&chemistry_configure` namelist (one derived-type variable `chemsol_params`, same
! component names as the reference's type_chemical_evol_solver_params, reference src/chemistry.f90:107-135,
! 183-184), the network / initial-abundance file formats and the cell-major, species-contiguous abundance
! layout; the work itself happens in libracgpu.so on the GPU.
module racgpu
  use, intrinsic :: iso_c_binding
  implicit none
  private
  public :: racgpu_params_t, type_chemical_evol_solver_params, chemsol_params, chemistry_configure_read
  public :: RACGPU_NPAR, RACGPU_NSTAT, RACGPU_NOUT, RACGPU_MEM_HOST, RACGPU_MEM_DEVICE, RACGPU_F_RECTIFY
  public :: racgpu_network_load, racgpu_network_destroy, racgpu_network_dims, racgpu_species_name, &
            racgpu_species_index, racgpu_load_initial_abundances, racgpu_params_default, racgpu_n_record, &
            racgpu_set_tolerances, racgpu_init_abundances, racgpu_set_device, racgpu_device_count, &
            racgpu_solve_batch, racgpu_evol_solve_batch, racgpu_calc_cells, racgpu_column_sweep, racgpu_set_co_shielding_table, racgpu_set_star_rays, racgpu_star_ray_timeouts, racgpu_rectify_abundances, &
            racgpu_set_cost_hints, racgpu_set_team_threshold, racgpu_rates, racgpu_last_error, racgpu_last_kernel_ms
  public :: racgpu_error_string, chemsol_to_c, c_string
  ! gas temperature co-evolving with the chemistry (evolT) and the single-process multi-GPU entry points
  public :: RACGPU_NHC, racgpu_hc_config_t, type_heating_cooling_config, heating_cooling_config, heating_cooling_configure_read, &
            heating_cooling_to_c, racgpu_hc_config_default, racgpu_heating_cooling_load, racgpu_evolT_solve_batch
  public :: racgpu_multi_create, racgpu_multi_destroy, racgpu_multi_calc_cells, racgpu_multi_error_string

  integer, parameter :: RACGPU_NPAR = 28, RACGPU_NSTAT = 20, RACGPU_NOUT = 6, RACGPU_MEM_HOST = 0, RACGPU_MEM_DEVICE = 1
  integer, parameter :: RACGPU_F_RECTIFY = 1
  integer, parameter :: RACGPU_NHC = 28 ! per-cell heating/cooling record (include/racgpu.h, RACGPU_H_*)

  ! struct racgpu_hc_config (include/racgpu.h)
  type, bind(c) :: racgpu_hc_config_t
    real(c_double) :: heating_eff_chem, heating_eff_H2form, heating_eff_phd_H2, heating_eff_phd_H2O, heating_eff_phd_OH, &
                      cooling_gg_coeff, base_alpha
    integer(c_int32_t) :: use_chemicalheatingcooling, use_Xray_heating, use_phdheating_H2, use_phdheating_H2OOH, &
                          use_mygasgraincooling, may_switch_T
  end type racgpu_hc_config_t

  ! The reference's &heating_cooling_configure namelist variable (src/heating_cooling.f90:16-38, 53-54): same component names and
  ! defaults, so that an existing configure file is accepted verbatim.  Branches this engine does not implement are refused.
  type :: type_heating_cooling_config
    logical :: use_analytical_CII_OI = .true.
    logical :: use_mygasgraincooling = .true.
    logical :: use_chemicalheatingcooling = .true.
    logical :: use_Xray_heating = .true.
    logical :: use_phdheating_H2 = .true.
    logical :: use_phdheating_H2OOH = .true.
    logical :: dust_gas_linear_couple = .false.
    integer :: solve_method = 1
    double precision :: heating_eff_chem = 1D0
    double precision :: heating_eff_H2form = 0.1D0
    double precision :: heating_eff_phd_H2 = 1D0
    double precision :: heating_eff_phd_H2O = 0.1D0
    double precision :: heating_eff_phd_OH = 0.1D0
    double precision :: heating_Xray_en = 18.7D0
    double precision :: cooling_gg_coeff = 0.3D0
    character(len=128) :: dir_transition_rates = './transitions/'
    character(len=128) :: filename_CII = 'C+.dat'
    character(len=128) :: filename_NII = 'N+.dat'
    character(len=128) :: filename_OI = 'Oatom.dat'
    character(len=128) :: filename_FeII = 'Fe+.dat'
    character(len=128) :: filename_SiII = 'Si+.dat'
    logical :: IonCoolingWithLut = .true.
  end type type_heating_cooling_config
  type(type_heating_cooling_config), save :: heating_cooling_config
  namelist /heating_cooling_configure/ heating_cooling_config

  ! struct racgpu_params (include/racgpu.h)
  type, bind(c) :: racgpu_params_t
    real(c_double) :: RTOL, ATOL, t_max, dt_first_step, ratio_tstep, max_runtime_allowed, Diff2DesorRatio, special_gH_E_diff
    integer(c_int32_t) :: mxstep_per_interval, steps_reset_solver, H2_form_use_moeq, evol_dust_size, &
                          use_special_gH_mobi, tol_policy_j
    integer(c_int64_t) :: max_steps_per_cell
    real(c_double) :: rt_cost_f, rt_cost_jac, rt_cost_lu
  end type racgpu_params_t

  ! The namelist variable.  Component names and defaults follow the reference's declaration so that an
  ! existing configure file is accepted verbatim; components the GPU path does not use are still parsed.
  type :: type_chemical_evol_solver_params
    character(len=128) :: chem_files_dir = './inp/', filename_chemical_network = '', &
                          filename_initial_abundances = '', filename_species_enthalpy = ''
    double precision :: RTOL = 1D-4, ATOL = 1D-30
    double precision :: t0 = 0D0, t_max = 1D6, t_max0 = 1D6, dt_first_step = 1D-6, dt_first_step0 = 0D0, &
                        ratio_tstep = 1.1D0, t_scale_tol = 0D0
    logical :: H2_form_use_moeq = .false.
    real :: max_runtime_allowed = 3600.0
    integer :: mxstep_per_interval = 2000
    integer :: n_record = 0, n_record_real = 0
    integer :: NEQ = 0, ITOL = 4, ITASK = 4, ISTATE = 1, IOPT = 1, LIW = 0, LRW = 0, MF = 21, NNZ = 0
    integer :: NERR = 0, quality = 0
    character(len=128) :: chem_evol_save_filename = 'chem_evol_tmp.dat'
    logical :: flag_chem_evol_save = .false.
    logical :: evolT = .false., maySwitchT = .false.
    logical :: evol_dust_size = .false.
    double precision :: Diff2DesorRatio = 0.5D0, Edesorb_gH_bare_grain = 1.0D4, Edesorb_gH_icy_grain = 450D0, &
                        special_gH_E_diff = 225D0
    logical :: update_gH_params_realtime = .false., use_special_gH_mobi = .false.
    integer :: steps_Update_gH_params = 5
    integer :: steps_reset_solver = 9999999
    integer :: fU_log = 6
  end type type_chemical_evol_solver_params

  type(type_chemical_evol_solver_params), save :: chemsol_params
  namelist /chemistry_configure/ chemsol_params

  interface
    function racgpu_last_error() bind(c, name='racgpu_last_error') result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
    function racgpu_device_count() bind(c, name='racgpu_device_count') result(n)
      import :: c_int
      integer(c_int) :: n
    end function
    function racgpu_set_device(dev) bind(c, name='racgpu_set_device') result(rc)
      import :: c_int
      integer(c_int), value :: dev
      integer(c_int) :: rc
    end function
    function racgpu_network_load(path) bind(c, name='racgpu_network_load') result(h)
      import :: c_ptr, c_char
      character(kind=c_char), dimension(*), intent(in) :: path
      type(c_ptr) :: h
    end function
    subroutine racgpu_network_destroy(h) bind(c, name='racgpu_network_destroy')
      import :: c_ptr
      type(c_ptr), value :: h
    end subroutine
    function racgpu_network_dims(h, nS, nR, nnzJ, nzl, nzu) bind(c, name='racgpu_network_dims') result(rc)
      import :: c_ptr, c_int, c_int32_t
      type(c_ptr), value :: h
      integer(c_int32_t), intent(out) :: nS, nR, nnzJ, nzl, nzu
      integer(c_int) :: rc
    end function
    function racgpu_species_name(h, i, buf, buflen) bind(c, name='racgpu_species_name') result(rc)
      import :: c_ptr, c_int, c_int32_t, c_char
      type(c_ptr), value :: h
      integer(c_int32_t), value :: i, buflen
      character(kind=c_char), dimension(*), intent(out) :: buf
      integer(c_int) :: rc
    end function
    function racgpu_species_index(h, name) bind(c, name='racgpu_species_index') result(i)
      import :: c_ptr, c_int, c_char
      type(c_ptr), value :: h
      character(kind=c_char), dimension(*), intent(in) :: name
      integer(c_int) :: i
    end function
    function racgpu_load_initial_abundances(h, path, y0) bind(c, name='racgpu_load_initial_abundances') result(rc)
      import :: c_ptr, c_int, c_char, c_double
      type(c_ptr), value :: h
      character(kind=c_char), dimension(*), intent(in) :: path
      real(c_double), dimension(*), intent(out) :: y0
      integer(c_int) :: rc
    end function
    subroutine racgpu_params_default(p) bind(c, name='racgpu_params_default')
      import :: racgpu_params_t
      type(racgpu_params_t), intent(out) :: p
    end subroutine
    function racgpu_n_record(p, t0, t_max) bind(c, name='racgpu_n_record') result(n)
      import :: racgpu_params_t, c_double, c_int
      type(racgpu_params_t), intent(in) :: p
      real(c_double), value :: t0, t_max
      integer(c_int) :: n
    end function
    function racgpu_set_tolerances(h, p, j, d2h, rtol, atol) bind(c, name='racgpu_set_tolerances') result(rc)
      import :: c_ptr, racgpu_params_t, c_int32_t, c_double, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int32_t), value :: j
      real(c_double), value :: d2h
      real(c_double), dimension(*), intent(out) :: rtol, atol
      integer(c_int) :: rc
    end function
    function racgpu_init_abundances(h, y0, cells, ncell, y) bind(c, name='racgpu_init_abundances') result(rc)
      import :: c_ptr, c_double, c_int64_t, c_int
      type(c_ptr), value :: h
      real(c_double), dimension(*), intent(in) :: y0, cells
      integer(c_int64_t), value :: ncell
      real(c_double), dimension(*), intent(out) :: y
      integer(c_int) :: rc
    end function
    function racgpu_rates(h, p, cells, ncell, rates) bind(c, name='racgpu_rates') result(rc)
      import :: c_ptr, racgpu_params_t, c_double, c_int64_t, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      real(c_double), dimension(*), intent(in) :: cells
      integer(c_int64_t), value :: ncell
      real(c_double), dimension(*), intent(out) :: rates
      integer(c_int) :: rc
    end function
    function racgpu_solve_batch(h, p, ncell, cells, y, t_final, quality, stats, record, touts, mem) &
        bind(c, name='racgpu_solve_batch') result(rc)
      import :: c_ptr, racgpu_params_t, c_int64_t, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int64_t), value :: ncell
      type(c_ptr), value :: cells, y, t_final, quality, stats, record, touts
      integer(c_int), value :: mem
      integer(c_int) :: rc
    end function
    ! chem_evol_solve per cell with its own t0 and tolerance policy j (continue runs; include/racgpu.h)
    function racgpu_evol_solve_batch(h, p, ncell, cells, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out, flags, mem) &
        bind(c, name='racgpu_evol_solve_batch') result(rc)
      import :: c_ptr, racgpu_params_t, c_int64_t, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int64_t), value :: ncell
      type(c_ptr), value :: cells, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out
      integer(c_int), value :: flags, mem
      integer(c_int) :: rc
    end function
    ! the local-iteration loop of calc_this_cell (reference src/disk.f90:1651-1791) for a batch of cells
    function racgpu_calc_cells(h, p, nlocal_iter, ncell, cells, y, t_final, quality, stats, cell_out, mem) &
        bind(c, name='racgpu_calc_cells') result(rc)
      import :: c_ptr, racgpu_params_t, c_int64_t, c_int32_t, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int32_t), value :: nlocal_iter
      integer(c_int64_t), value :: ncell
      type(c_ptr), value :: cells, y, t_final, quality, stats, cell_out
      integer(c_int), value :: mem
      integer(c_int) :: rc
    end function
    function racgpu_rectify_abundances(h, ncell, y) bind(c, name='racgpu_rectify_abundances') result(rc)
      import :: c_ptr, c_int64_t, c_double, c_int
      type(c_ptr), value :: h
      integer(c_int64_t), value :: ncell
      real(c_double), dimension(*), intent(inout) :: y
      integer(c_int) :: rc
    end function
    ! scheduling hint for the following racgpu_solve_batch calls: expected work per cell, costliest cells first
    function racgpu_set_cost_hints(h, cost, ncell) bind(c, name='racgpu_set_cost_hints') result(rc)
      import :: c_ptr, c_double, c_int64_t, c_int
      type(c_ptr), value :: h
      real(c_double), dimension(*), intent(in) :: cost
      integer(c_int64_t), value :: ncell
      integer(c_int) :: rc
    end function
    ! 12CO shielding table for racgpu_column_sweep: f(ncol, nrow) over ascending logN_12CO(ncol), logN_H2(nrow)
    function racgpu_set_co_shielding_table(h, nrow, ncol, logN_H2, logN_12CO, f) bind(c, name='racgpu_set_co_shielding_table') result(rc)
      import :: c_ptr, c_int32_t, c_double, c_int
      type(c_ptr), value :: h
      integer(c_int32_t), value :: nrow, ncol
      real(c_double), dimension(*), intent(in) :: logN_H2, logN_12CO, f
      integer(c_int) :: rc
    end function
    ! rays to the star for racgpu_column_sweep (calc_Ncol_to_Star, src/disk.f90:2543-2555, in the one-predecessor form of a column grid):
    ! inner(cell) = the 0-based cell the ray from `cell` enters next (-1: none), ds(cell) = path length of such a ray through `cell` [cm];
    ! with rays set the sweep also rewrites the toStar slots and a cell waits for inner(cell).  ncell = 0 or a null inner clears.
    function racgpu_set_star_rays(h, ncell, inner, ds) bind(c, name='racgpu_set_star_rays') result(rc)
      import :: c_ptr, c_int64_t, c_int32_t, c_double, c_int
      type(c_ptr), value :: h
      integer(c_int64_t), value :: ncell
      integer(c_int32_t), dimension(*), intent(in) :: inner
      real(c_double), dimension(*), intent(in) :: ds
      integer(c_int) :: rc
    end function
    function racgpu_star_ray_timeouts(h) bind(c, name='racgpu_star_ray_timeouts') result(n)
      import :: c_ptr, c_int
      type(c_ptr), value :: h
      integer(c_int) :: n
    end function
    ! the sweep in dependency order for grids whose cells form columns (include/racgpu.h): columns top down, the toISM self-shielding
    ! slots of H2, H2O and OH rewritten on the device from the cells above; col_ptr/col_cells are 0-based
    function racgpu_column_sweep(h, p, ncolumn, col_ptr, col_cells, ncell, cells, y, dz, dv_turb, t_final, quality, stats, &
                                 cell_out, mem) bind(c, name='racgpu_column_sweep') result(rc)
      import :: c_ptr, c_int64_t, c_int32_t, c_double, c_int, racgpu_params_t
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int64_t), value :: ncolumn, ncell
      integer(c_int32_t), dimension(*), intent(in) :: col_ptr, col_cells
      real(c_double), dimension(*), intent(inout) :: cells, y
      real(c_double), dimension(*), intent(in) :: dz
      real(c_double), value :: dv_turb
      real(c_double), dimension(*), intent(out) :: t_final, cell_out
      integer(c_int32_t), dimension(*), intent(out) :: quality
      integer(c_int64_t), dimension(*), intent(out) :: stats
      integer(c_int), value :: mem
      integer(c_int) :: rc
    end function
    ! cells expected to cost more than frac x (sum of the hints / wave slots) get a team of four waves (default 0.5; <= 0 never;
    ! < 0 also switches off the hand-over of the last running cells to teams at the end of a pass); results do not depend on it
    function racgpu_set_team_threshold(h, frac) bind(c, name='racgpu_set_team_threshold') result(rc)
      import :: c_ptr, c_double, c_int
      type(c_ptr), value :: h
      real(c_double), value :: frac
      integer(c_int) :: rc
    end function
    function racgpu_last_kernel_ms(h) bind(c, name='racgpu_last_kernel_ms') result(ms)
      import :: c_ptr, c_double
      type(c_ptr), value :: h
      real(c_double) :: ms
    end function
    subroutine racgpu_hc_config_default(c) bind(c, name='racgpu_hc_config_default')
      import :: racgpu_hc_config_t
      type(racgpu_hc_config_t), intent(out) :: c
    end subroutine
    ! heating_cooling_prepare + the reaction heats: enthalpy file, Neufeld tables (compiled into the reference; here a data file), ion LUTs
    function racgpu_heating_cooling_load(h, c, enthalpy_file, neufeld_tables, nii_lut, siii_lut, feii_lut) &
        bind(c, name='racgpu_heating_cooling_load') result(rc)
      import :: c_ptr, racgpu_hc_config_t, c_char, c_int
      type(c_ptr), value :: h
      type(racgpu_hc_config_t), intent(in) :: c
      character(kind=c_char), dimension(*), intent(in) :: enthalpy_file, neufeld_tables, nii_lut, siii_lut, feii_lut
      integer(c_int) :: rc
    end function
    ! racgpu_evol_solve_batch with the gas temperature co-evolving (chemsol_params%evolT): hc = [RACGPU_NHC, ncell]
    function racgpu_evolT_solve_batch(h, p, ncell, cells, hc, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out, flags, mem) &
        bind(c, name='racgpu_evolT_solve_batch') result(rc)
      import :: c_ptr, racgpu_params_t, c_int64_t, c_int
      type(c_ptr), value :: h
      type(racgpu_params_t), intent(in) :: p
      integer(c_int64_t), value :: ncell
      type(c_ptr), value :: cells, hc, y, t0, tol_j, t_final, quality, stats, record, touts, cell_out
      integer(c_int), value :: flags, mem
      integer(c_int) :: rc
    end function
    ! one process, ndev GPUs, one RCCL all-gather of the results (include/racgpu.h); devices = c_null_ptr: 0 .. ndev-1
    function racgpu_multi_create(path, ndev, devices) bind(c, name='racgpu_multi_create') result(m)
      import :: c_ptr, c_char, c_int
      character(kind=c_char), dimension(*), intent(in) :: path
      integer(c_int), value :: ndev
      type(c_ptr), value :: devices
      type(c_ptr) :: m
    end function
    subroutine racgpu_multi_destroy(m) bind(c, name='racgpu_multi_destroy')
      import :: c_ptr
      type(c_ptr), value :: m
    end subroutine
    function racgpu_multi_last_error() bind(c, name='racgpu_multi_last_error') result(p)
      import :: c_ptr
      type(c_ptr) :: p
    end function
    function racgpu_multi_calc_cells(m, p, nlocal_iter, ncell, cells, y, t_final, quality, stats, cell_out, cost) &
        bind(c, name='racgpu_multi_calc_cells') result(rc)
      import :: c_ptr, racgpu_params_t, c_int64_t, c_int32_t, c_int
      type(c_ptr), value :: m
      type(racgpu_params_t), intent(in) :: p
      integer(c_int32_t), value :: nlocal_iter
      integer(c_int64_t), value :: ncell
      type(c_ptr), value :: cells, y, t_final, quality, stats, cell_out, cost
      integer(c_int) :: rc
    end function
  end interface

contains

  function c_string(s) result(c)
    character(len=*), intent(in) :: s
    character(kind=c_char), dimension(len_trim(s) + 1) :: c
    integer :: i
    do i = 1, len_trim(s)
      c(i) = s(i:i)
    end do
    c(len_trim(s) + 1) = c_null_char
  end function c_string

  function racgpu_error_string() result(s)
    character(len=256) :: s
    character(kind=c_char), dimension(:), pointer :: p
    type(c_ptr) :: cp
    integer :: i
    s = ''
    cp = racgpu_last_error()
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [256])
    do i = 1, 256
      if (p(i) == c_null_char) exit
      s(i:i) = p(i)
    end do
  end function racgpu_error_string

  function racgpu_multi_error_string() result(s)
    character(len=256) :: s
    character(kind=c_char), dimension(:), pointer :: p
    type(c_ptr) :: cp
    integer :: i
    s = ''
    cp = racgpu_multi_last_error()
    if (.not. c_associated(cp)) return
    call c_f_pointer(cp, p, [256])
    do i = 1, 256
      if (p(i) == c_null_char) exit
      s(i:i) = p(i)
    end do
  end function racgpu_multi_error_string

  ! read &heating_cooling_configure as the reference does (src/configure.f90:29); ios /= 0 when the file has none
  subroutine heating_cooling_configure_read(funit, ios)
    integer, intent(in) :: funit
    integer, intent(out) :: ios
    read(funit, nml=heating_cooling_configure, iostat=ios)
  end subroutine heating_cooling_configure_read

  ! namelist values -> the C struct; the branches the engine does not implement are refused, not ignored
  subroutine heating_cooling_to_c(c, base_alpha, may_switch_T)
    type(racgpu_hc_config_t), intent(out) :: c
    double precision, intent(in) :: base_alpha
    logical, intent(in) :: may_switch_T
    if ((.not. heating_cooling_config%use_analytical_CII_OI) .or. (.not. heating_cooling_config%IonCoolingWithLut) .or. &
        heating_cooling_config%dust_gas_linear_couple) then
      write(*, '(A)') 'racgpu: only use_analytical_CII_OI = IonCoolingWithLut = .true., dust_gas_linear_couple = .false. are implemented'
      stop 1
    end if
    call racgpu_hc_config_default(c)
    c%heating_eff_chem = heating_cooling_config%heating_eff_chem
    c%heating_eff_H2form = heating_cooling_config%heating_eff_H2form
    c%heating_eff_phd_H2 = heating_cooling_config%heating_eff_phd_H2
    c%heating_eff_phd_H2O = heating_cooling_config%heating_eff_phd_H2O
    c%heating_eff_phd_OH = heating_cooling_config%heating_eff_phd_OH
    c%cooling_gg_coeff = heating_cooling_config%cooling_gg_coeff
    c%base_alpha = base_alpha
    c%use_chemicalheatingcooling = merge(1, 0, heating_cooling_config%use_chemicalheatingcooling)
    c%use_Xray_heating = merge(1, 0, heating_cooling_config%use_Xray_heating)
    c%use_phdheating_H2 = merge(1, 0, heating_cooling_config%use_phdheating_H2)
    c%use_phdheating_H2OOH = merge(1, 0, heating_cooling_config%use_phdheating_H2OOH)
    c%use_mygasgraincooling = merge(1, 0, heating_cooling_config%use_mygasgraincooling)
    c%may_switch_T = merge(1, 0, may_switch_T)
  end subroutine heating_cooling_to_c

  ! read &chemistry_configure exactly as the reference does (src/configure.f90:28)
  subroutine chemistry_configure_read(funit, ios)
    integer, intent(in) :: funit
    integer, intent(out) :: ios
    read(funit, nml=chemistry_configure, iostat=ios)
  end subroutine chemistry_configure_read

  ! namelist values -> the C struct
  subroutine chemsol_to_c(p)
    type(racgpu_params_t), intent(out) :: p
    ! switches of the reference this engine does not implement are refused, not ignored (DESIGN.md section 0)
    if (chemsol_params%update_gH_params_realtime) then
      write(*, '(A)') 'racgpu: chemsol_params%update_gH_params_realtime = .true. is not implemented'
      stop 1
    end if
    call racgpu_params_default(p)
    p%RTOL = chemsol_params%RTOL
    p%ATOL = chemsol_params%ATOL
    p%t_max = chemsol_params%t_max
    p%dt_first_step = chemsol_params%dt_first_step
    p%ratio_tstep = chemsol_params%ratio_tstep
    p%max_runtime_allowed = dble(chemsol_params%max_runtime_allowed)
    p%Diff2DesorRatio = chemsol_params%Diff2DesorRatio
    p%special_gH_E_diff = chemsol_params%special_gH_E_diff
    p%mxstep_per_interval = chemsol_params%mxstep_per_interval
    p%steps_reset_solver = chemsol_params%steps_reset_solver
    p%H2_form_use_moeq = merge(1, 0, chemsol_params%H2_form_use_moeq)
    p%evol_dust_size = merge(1, 0, chemsol_params%evol_dust_size)
    p%use_special_gH_mobi = merge(1, 0, chemsol_params%use_special_gH_mobi)
    p%tol_policy_j = 1
    p%max_steps_per_cell = 0
  end subroutine chemsol_to_c

end module racgpu
